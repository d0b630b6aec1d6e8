"""The index arithmetic of the fused ViT front end (trace_amd/csrc/patch_embed.hip), restated in numpy and checked against the convolution it replaces
(HF CLIPVisionEmbeddings: Conv2d(3, D, kernel = stride = P, bias = False), reached from trace/model/multimodal_encoder/clip_encoder.py:50):
  * k is re-indexed kk = (c P + ky) 16 + j with j padded P -> 16; the weight is repacked into that order with zeros in the pad columns (patch_pack_kernel);
  * a lane's fragment = 8 consecutive pixels of one frame row starting at pixel gx P + {0, 8}; on the last patch column the second window would run
    past the row: it is read `ov` pixels earlier and shifted; the pad positions are zeroed.
CPU only — the mapping the kernel header describes, not the kernel (tests/test_gpu_parity.py::test_fused_patch_embed_matches_three_pass_front_end runs that)."""
import numpy as np
import pytest
import torch


def pack_weight(W, P):
    """[D, 3 P P] (k = c P P + ky P + j) -> [D, 3 P 16] (kk = (c P + ky) 16 + j), zero for j >= P"""
    D = W.shape[0]
    out = np.zeros((D, 3 * P * 16), W.dtype)
    for c in range(3):
        for ky in range(P):
            out[:, (c * P + ky) * 16:(c * P + ky) * 16 + P] = W[:, c * P * P + ky * P: c * P * P + ky * P + P]
    return out


def patch_row(frame, gy, gx, P):
    """the kernel's A row of patch (gy, gx): for every (c, ky) pair two 8-pixel windows, loaded the way xload() does"""
    S = frame.shape[-1]
    flat = frame.reshape(-1)                                    # [3 S S]: a window that starts early stays inside the frame's memory
    row = np.zeros(3 * P * 16, frame.dtype)
    for c in range(3):
        for ky in range(P):
            for half in (0, 1):
                j0 = half * 8
                ov = max(0, gx * P + j0 + 8 - S)
                off = (c * S + gy * P + ky) * S + gx * P + j0 - ov
                v = flat[off:off + 8].copy()
                if ov:                                          # shift left by ov elements, zeros behind
                    v = np.concatenate([v[ov:], np.zeros(ov, v.dtype)])
                if j0:
                    v[max(0, P - 8):] = 0                       # pad columns j >= P
                row[(c * P + ky) * 16 + j0:(c * P + ky) * 16 + j0 + 8] = v
    return row


@pytest.mark.parametrize("P,G", [(14, 3), (14, 1), (16, 2)])
def test_reindexed_patch_gemm_equals_the_convolution(P, G):
    rng = np.random.RandomState(P * 10 + G)
    S, D = P * G, 24
    frame = rng.randn(3, S, S)
    W = rng.randn(D, 3 * P * P)
    Wp = pack_weight(W, P)
    assert (3 * P * 16) % 32 == 0                               # whole MFMA k-steps (21 for P = 14)
    want = torch.nn.functional.conv2d(torch.from_numpy(frame)[None], torch.from_numpy(W).view(D, 3, P, P), stride=P)[0].numpy()      # [D, G, G]
    for gy in range(G):
        for gx in range(G):
            got = Wp @ patch_row(frame, gy, gx, P)
            np.testing.assert_allclose(got, want[:, gy, gx], rtol=1e-10, atol=1e-10)


def test_a_nan_pixel_stays_in_its_own_patch():
    """the second window of a patch overlaps the first two pixels of its right neighbour (14 -> 16 padding): those positions are zeroed, so a
    non-finite pixel cannot leak into the neighbouring patch through 0 x NaN"""
    P, G = 14, 2
    frame = np.zeros((3, P * G, P * G))
    frame[0, 0, P] = np.nan                                      # first pixel of patch (0, 1)
    assert np.isfinite(patch_row(frame, 0, 0, P)).all()
    assert not np.isfinite(patch_row(frame, 0, 1, P)).all()

"""Container readers of trace_amd/video_io.py (`process_video` from a FILE without decord): YUV4MPEG2 and AVI (Motion-JPEG /
uncompressed) through the same sampling + timestamp code as decord would feed (trace/mm_utils.py:421-437).  CPU only."""
import numpy as np
import pytest

from trace_amd import video_io as vio
from trace_amd.mm_utils import process_video, sample_indices_and_timestamps


def _clip(T=40, H=48, W=64, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(1, (H + 7) // 8, (W + 7) // 8, 3)).repeat(8, axis=1).repeat(8, axis=2)[:, :H, :W]       # 8x8 blocks: chroma subsampling is exact on them
    drift = (np.arange(T)[:, None, None, None] * 3) % 64
    return ((base.astype(np.int64) + drift) % 256).astype(np.uint8)


def test_yuv_formulas_against_float_reference():
    """the integer BT.601 limited-range conversion equals the rounded floating-point formulas of the docstrings, on every (Y, U, V) of a grid"""
    y, u, v = np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 7), np.arange(0, 256, 7), indexing="ij")
    got = vio.yuv601_to_rgb(y.astype(np.uint8), u.astype(np.uint8), v.astype(np.uint8)).astype(np.int64)
    c, d, e = y - 16.0, u - 128.0, v - 128.0
    want = np.stack([1.164383 * c + 1.596027 * e, 1.164383 * c - 0.391762 * d - 0.812968 * e, 1.164383 * c + 2.017232 * d], -1)
    want = np.clip(np.floor(want + 0.5), 0, 255)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01          # equal up to ties of the fixed-point rounding
    rgb = np.random.RandomState(1).randint(0, 256, size=(64, 64, 3)).astype(np.uint8)
    back = vio.yuv601_to_rgb(*vio.rgb_to_yuv601(rgb)).astype(int)
    assert np.abs(back - rgb.astype(int)).max() <= 3                              # 8-bit limited-range round trip


@pytest.mark.parametrize("chroma", ["420", "422", "444", "mono"])
def test_y4m_round_trip(tmp_path, chroma):
    fr = _clip(T=12, H=46, W=62)                                                  # odd chroma sizes on purpose (46 / 2 = 23, 62 / 2 = 31)
    p = str(tmp_path / f"c_{chroma}.y4m")
    vio.write_y4m(p, fr, fps=(30000, 1001), chroma=chroma)
    vr = vio.Y4MReader(p)
    assert len(vr) == 12 and abs(vr.get_avg_fps() - 29.97) < 1e-2 and (vr.width, vr.height) == (62, 46)
    got = vr.get_batch([0, 5, 11]).asnumpy()
    assert got.shape == (3, 46, 62, 3) and got.dtype == np.uint8
    if chroma == "444":
        assert np.abs(got.astype(int) - fr[[0, 5, 11]].astype(int)).max() <= 3
    if chroma == "mono":
        assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()
    assert np.array_equal(vr[5], got[1])
    with pytest.raises(IndexError):
        vr.get_batch([12])


def test_y4m_block_clip_is_exact_under_subsampling(tmp_path):
    fr = _clip(T=4)
    p444, p420 = str(tmp_path / "a.y4m"), str(tmp_path / "b.y4m")
    vio.write_y4m(p444, fr, chroma="444")
    vio.write_y4m(p420, fr, chroma="420")
    assert np.array_equal(vio.Y4MReader(p444).get_batch(range(4)).asnumpy(), vio.Y4MReader(p420).get_batch(range(4)).asnumpy())


def test_y4m_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.y4m"
    p.write_bytes(b"YUV4MPEG2 W4 H4 F25:1 C420p10\nFRAME\n" + bytes(48))
    with pytest.raises(ValueError, match="unsupported y4m colour space"):
        vio.Y4MReader(str(p))
    p.write_bytes(b"RIFFxxxx")
    with pytest.raises(ValueError, match="not a YUV4MPEG2"):
        vio.Y4MReader(str(p))
    p.write_bytes(b"YUV4MPEG2 W4 H4 F25:1 C444\nFRAME\n" + bytes(48) + b"FRAME Ip\n" + bytes(48) + b"FRAME\n" + bytes(20))      # frame parameters; truncated tail
    assert len(vio.Y4MReader(str(p))) == 2


@pytest.mark.parametrize("codec", ["MJPG", "DIB "])
def test_avi_round_trip(tmp_path, codec):
    fr = _clip(T=9, H=47, W=63)
    p = str(tmp_path / "c.avi")
    vio.write_avi(p, fr, fps=(12, 1), codec=codec)
    vr = vio.AviReader(p)
    assert len(vr) == 9 and vr.get_avg_fps() == 12.0 and (vr.width, vr.height) == (63, 47)
    got = vr.get_batch([8, 0]).asnumpy()
    if codec == "DIB ":
        assert np.array_equal(got, fr[[8, 0]])
    else:
        assert np.abs(got.astype(int) - fr[[8, 0]].astype(int)).mean() < 2.0     # JPEG at quality 95, 4:4:4


def test_avi_refuses_inter_coded_streams(tmp_path):
    fr = _clip(T=2)
    p = str(tmp_path / "h.avi")
    vio.write_avi(p, fr, codec="DIB ")
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"strf") + 8 + 16
    raw[i:i + 4] = b"H264"
    j = raw.index(b"vids") + 4
    raw[j:j + 4] = b"H264"
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="inter-coded"):
        vio.AviReader(p)


class _Proc:
    """stand-in for the HF image processor: records the PIL frames it is handed"""
    image_mean = [0.48145466, 0.4578275, 0.40821073]

    def preprocess(self, images, return_tensors=None):
        import torch
        self.seen = [np.asarray(im) for im in images]
        return {"pixel_values": torch.zeros(len(images), 3, 4, 4)}


@pytest.mark.parametrize("ext", ["y4m", "avi"])
def test_process_video_from_a_file_equals_process_video_from_its_frames(tmp_path, ext):
    """the reference's decord branch (mm_utils.py:421-437) over these readers: same indices, same timestamps (index / fps of the FILE), same frames
    as handing the decoded array to process_video with that fps"""
    fr = _clip(T=40)
    p = str(tmp_path / f"clip.{ext}")
    if ext == "y4m":
        vio.write_y4m(p, fr, fps=(8, 1), chroma="444")
    else:
        vio.write_avi(p, fr, fps=(8, 1), codec="DIB ")
    decoded = vio.open_container(p).get_batch(range(40)).asnumpy()
    a, b = _Proc(), _Proc()
    _, ts_file = process_video(p, a, "pad", 8)
    _, ts_arr = process_video(decoded, b, "pad", 8, fps=8.0)
    idx, ts_want = sample_indices_and_timestamps(40, 8.0, 8)
    assert ts_file == ts_arr == ts_want and list(idx) == [0, 5, 11, 16, 22, 27, 33, 39]
    assert len(a.seen) == 8 and all(np.array_equal(x, y) for x, y in zip(a.seen, b.seen))
    assert a.seen[0].shape == (64, 64, 3)                                         # expand2square of 48 x 64 frames


def test_mp4_with_still_image_samples_round_trip(tmp_path):
    """ISO-BMFF with a Photo-JPEG / Motion-JPEG video track: the sample table (stsd / stts / stsc / stsz / stco) is parsed here, the samples go to Pillow"""
    fr = _clip(T=11, H=46, W=62)
    p = str(tmp_path / "c.mp4")
    vio.write_mjpeg_mp4(p, fr, fps=(30000, 1001))
    vr = vio.open_container(p)
    assert isinstance(vr, vio.Mp4Reader) and len(vr) == 11 and abs(vr.get_avg_fps() - 29.97) < 1e-2 and (vr.width, vr.height) == (62, 46)
    got = vr.get_batch([10, 0, 5]).asnumpy()
    assert got.shape == (3, 46, 62, 3) and np.abs(got.astype(int) - fr[[10, 0, 5]].astype(int)).mean() < 2.0
    with pytest.raises(IndexError):
        vr.get_batch([11])
    a, b = _Proc(), _Proc()
    _, ts_file = process_video(p, a, "pad", 4)
    _, ts_arr = process_video(vr.get_batch(range(11)).asnumpy(), b, "pad", 4, fps=vr.get_avg_fps())
    assert ts_file == ts_arr and all(np.array_equal(x, y) for x, y in zip(a.seen, b.seen))


def test_containers_that_need_a_codec_say_so(tmp_path):
    """an inter-coded track is named by its fourcc and handed to decord as the reference does (mm_utils.py:421) — without decord: ImportError saying so;
    something the parsers cannot make sense of goes the same way (round 5: the readers are a fast path, never a gate) and reads "not readable here"
    when decord is absent"""
    fr = _clip(T=3)
    p = str(tmp_path / "clip.mp4")
    vio.write_mjpeg_mp4(p, fr, fourcc=b"avc1")
    with pytest.raises(ValueError, match="avc1"):
        vio.Mp4Reader(p)
    try:
        import decord  # noqa: F401
        has_decord = True
    except ImportError:
        has_decord = False
    if not has_decord:
        with pytest.raises(ImportError, match="decord"):
            process_video(p, _Proc(), "pad", 8)
    q = tmp_path / "junk.mp4"
    q.write_bytes(b"\0\0\0\x18ftypisom")
    with pytest.raises(ValueError, match="not an ISO base media file"):
        vio.Mp4Reader(str(q))
    if not has_decord:
        with pytest.raises(ImportError, match="not an ISO base media file"):
            process_video(str(q), _Proc(), "pad", 8)
    w = tmp_path / "clip.webm"
    w.write_bytes(b"\x1aE\xdf\xa3")
    if not has_decord:
        with pytest.raises(ImportError, match="decord"):
            process_video(str(w), _Proc(), "pad", 8)


def test_a_reader_that_fails_while_parsing_leaves_no_descriptor_behind(tmp_path):
    """A stream header cut short makes AviReader raise from inside its parse (struct.error): the map, the view and the file must be closed by then —
    open_container() hands such files to decord, and a long evaluation would otherwise run out of descriptors."""
    import os, struct
    strh = b"strh" + struct.pack("<I", 56) + b"vids" + b"MJPG" + b"\0" * 8            # 'vids' header whose scale / rate fields lie past the end of the file
    strl = b"LIST" + struct.pack("<I", 4 + len(strh)) + b"strl" + strh
    hdrl = b"LIST" + struct.pack("<I", 4 + len(strl)) + b"hdrl" + strl
    body = b"AVI " + hdrl
    p = tmp_path / "cut.avi"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    before = len(os.listdir("/proc/self/fd"))
    for _ in range(4):
        with pytest.raises((ValueError, struct.error)):
            vio.AviReader(str(p))
    assert len(os.listdir("/proc/self/fd")) == before
    # a well-formed header with no frames takes the other exit (ValueError after the parse): closed as well
    q = tmp_path / "empty.avi"
    q.write_bytes(b"RIFF" + struct.pack("<I", 4) + b"AVI ")
    for _ in range(4):
        with pytest.raises((ValueError, vio.NeedsDecoder)):
            vio.AviReader(str(q))
    assert len(os.listdir("/proc/self/fd")) == before

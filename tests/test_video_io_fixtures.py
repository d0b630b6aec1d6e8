"""Container readers (trace_amd/video_io.py) against bytes this repo's writers did NOT produce: minimal AVI (RIFF / hdrl / movi / idx1)
and ISO-BMFF (ftyp / moov / stbl) files assembled here field by field from the format specifications — Microsoft's "AVI RIFF File Reference" and
OpenDML 1.02 for the chunk, header and index layouts; ISO/IEC 14496-12 for the boxes — with the features a real muxer emits and write_avi /
write_mjpeg_mp4 never do: JUNK padding, an audio stream interleaved in 'movi', odd-sized chunks, 'rec ' groups, an 'AVIX' continuation segment, dropped
(zero-length) frames, an index; 'moov' in front of 'mdat', a version-1 'mdhd', several samples per chunk with a multi-entry 'stsc', 64-bit chunk offsets
('co64'), a 64-bit 'mdat' header, a 'free' box, a second (audio) track in front of the video track.  The helpers below share no code with video_io's
writers (they do not import them).  CPU only."""
import io
import os
import struct

import numpy as np
import pytest

from trace_amd import video_io as V


def _frames(T=7, H=10, W=14, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(H, W, 3))
    return np.stack([(base + 17 * t) % 256 for t in range(T)]).astype(np.uint8)


# ------------------------------------------------------------------------------------------------------------ AVI, from the RIFF specification
def _ck(fourcc, payload):
    assert len(fourcc) == 4
    return fourcc + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) % 2 else b"")


def _list(kind, body):
    return b"LIST" + struct.pack("<I", 4 + len(body)) + kind + body


def _dib_bytes(rgb):
    """one uncompressed 24-bit frame as a bottom-up DIB: rows bottom to top, pixels B G R, every row padded to a multiple of 4 bytes"""
    H, W, _ = rgb.shape
    out = bytearray()
    for y in range(H - 1, -1, -1):
        row = bytearray()
        for x in range(W):
            r, g, b = (int(c) for c in rgb[y, x])
            row += bytes((b, g, r))
        row += b"\x00" * ((-len(row)) % 4)
        out += row
    return bytes(out)


def _avi_headers(T, H, W, rate, scale, with_audio):
    usec = int(round(1e6 * scale / rate))
    frame_bytes = ((W * 3 + 3) // 4 * 4) * H
    avih = struct.pack("<IIIIIIIIII", usec, 0, 0, 0x10 | 0x100, T, 0, 2 if with_audio else 1, frame_bytes, W, H) + bytes(16)      # AVIF_HASINDEX | AVIF_ISINTERLEAVED
    strh_v = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, scale, rate, 0, T, frame_bytes, 0xFFFFFFFF, 0) + struct.pack("<hhhh", 0, 0, W, H)
    bih = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, frame_bytes, 2835, 2835, 0, 0)          # BITMAPINFOHEADER, BI_RGB, positive height = bottom-up
    strl = _list(b"strl", _ck(b"strh", strh_v) + _ck(b"strf", bih) + _ck(b"strn", b"video\x00"))
    if with_audio:
        strh_a = b"auds" + bytes(4) + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, 8000, 0, 8000, 4096, 0xFFFFFFFF, 1) + bytes(8)
        wfx = struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8)                                       # WAVEFORMATEX (PCM, mono, 8 kHz, 8 bit) without cbSize
        strl = _list(b"strl", _ck(b"strh", strh_a) + _ck(b"strf", wfx)) + strl                     # the audio stream FIRST: the video stream is number 01
    return _list(b"hdrl", _ck(b"avih", avih) + strl)


def test_avi_assembled_from_the_riff_spec(tmp_path):
    fr = _frames()
    T, H, W, _ = fr.shape
    vid = b"01db"                                                     # stream 01 = video (stream 00 is the audio stream)
    body, index, pos = b"", b"", 4                                    # idx1 offsets count from the 'movi' fourcc
    for t in range(T):
        for cc, payload in ((b"00wb", bytes(range(33))), (vid, _dib_bytes(fr[t]))):       # a 33-byte (odd) audio chunk in front of every frame
            c = _ck(cc, payload)
            index += cc + struct.pack("<III", 0x10, pos, len(payload))
            body += c
            pos += len(c)
    riff = b"AVI " + _avi_headers(T, H, W, 30000, 1001, True) + _ck(b"JUNK", bytes(57)) + _list(b"movi", body) + _ck(b"idx1", index)
    p = str(tmp_path / "spec.avi")
    with open(p, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff)) + riff)
    r = V.open_container(p)
    assert isinstance(r, V.AviReader) and len(r) == T and (r.height, r.width) == (H, W)
    assert abs(r.get_avg_fps() - 30000 / 1001) < 1e-9
    got = r.get_batch(list(range(T))).asnumpy()
    assert np.array_equal(got, fr)
    assert np.array_equal(r[3], fr[3])


def test_avi_opendml_segments_groups_and_dropped_frames(tmp_path):
    """OpenDML: the 'movi' list of the first RIFF chunk groups its chunks in 'rec ' lists, a second 'RIFF....AVIX' chunk carries more frames, and a
    zero-length video chunk is a dropped frame (the previous picture is shown again and the time line keeps its place)."""
    fr = _frames(T=6)
    T, H, W, _ = fr.shape
    seg1 = _list(b"rec ", _ck(b"00db", _dib_bytes(fr[0])) + _ck(b"00db", _dib_bytes(fr[1]))) + \
        _list(b"rec ", _ck(b"00db", b"") + _ck(b"00db", _dib_bytes(fr[2])))                      # frame 2 of the file = a dropped frame (repeats fr[1])
    seg2 = _ck(b"00db", _dib_bytes(fr[3])) + _ck(b"00db", _dib_bytes(fr[4])) + _ck(b"00db", _dib_bytes(fr[5]))
    riff1 = b"AVI " + _avi_headers(7, H, W, 25, 1, False) + _list(b"movi", seg1)
    riff2 = b"AVIX" + _list(b"movi", seg2)
    p = str(tmp_path / "odml.avi")
    with open(p, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff1)) + riff1 + b"RIFF" + struct.pack("<I", len(riff2)) + riff2)
    r = V.open_container(p)
    assert len(r) == 7
    want = np.stack([fr[0], fr[1], fr[1], fr[2], fr[3], fr[4], fr[5]])
    assert np.array_equal(r.get_batch(list(range(7))).asnumpy(), want)


def test_avi_with_an_inter_coded_stream_goes_to_decord(tmp_path):
    """XVID / H.264 in AVI is common in video datasets: the reference hands every file to decord (mm_utils.py:421), so must this build — the reader is a
    fast path, not a gate.  Without decord the message names the codec."""
    H, W = 8, 8
    avih = struct.pack("<IIIIIIIIII", 40000, 0, 0, 0x10, 1, 0, 1, 0, W, H) + bytes(16)
    strh = b"vids" + b"XVID" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, 25, 0, 1, 0, 0xFFFFFFFF, 0) + struct.pack("<hhhh", 0, 0, W, H)
    bih = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"XVID", 0, 0, 0, 0, 0)
    riff = b"AVI " + _list(b"hdrl", _ck(b"avih", avih) + _list(b"strl", _ck(b"strh", strh) + _ck(b"strf", bih))) + _list(b"movi", _ck(b"00dc", bytes(40)))
    p = str(tmp_path / "xvid.avi")
    with open(p, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff)) + riff)
    with pytest.raises(V.NeedsDecoder, match="XVID"):
        V.AviReader(p)
    try:
        import decord  # noqa: F401
        assert V.open_container(p) is None
    except ImportError:
        with pytest.raises(ImportError, match="XVID"):
            V.open_container(p)


# ------------------------------------------------------------------------------------------------------------ ISO-BMFF, from ISO/IEC 14496-12
def _box(typ, payload, large=False):
    if large:                                                         # size == 1: a 64-bit `largesize` follows the type
        return struct.pack(">I4sQ", 1, typ, 16 + len(payload)) + payload
    return struct.pack(">I4s", 8 + len(payload), typ) + payload


def _fullbox(typ, version, flags, payload):
    return _box(typ, struct.pack(">B", version) + flags.to_bytes(3, "big") + payload)


def _png(rgb):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, format="PNG")
    return b.getvalue()


_UNITY = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def _trak(handler, stsd_entry, samples_per_chunk_runs, sizes, chunk_offsets, timescale, deltas, W, H, track_id):
    T = len(sizes)
    dur = sum(n * d for n, d in deltas)
    tkhd = _fullbox(b"tkhd", 0, 7, struct.pack(">IIIII", 0, 0, track_id, 0, dur) + bytes(8) + struct.pack(">hhhH", 0, 0, 0, 0) + _UNITY + struct.pack(">II", W << 16, H << 16))
    mdhd = _fullbox(b"mdhd", 1, 0, struct.pack(">QQIQ", 0, 0, timescale, dur) + struct.pack(">HH", 0x55C4, 0))           # version 1: 64-bit times
    hdlr = _fullbox(b"hdlr", 0, 0, bytes(4) + handler + bytes(12) + b"handler\x00")
    stsd = _fullbox(b"stsd", 0, 0, struct.pack(">I", 1) + stsd_entry)
    stts = _fullbox(b"stts", 0, 0, struct.pack(">I", len(deltas)) + b"".join(struct.pack(">II", n, d) for n, d in deltas))
    stsc = _fullbox(b"stsc", 0, 0, struct.pack(">I", len(samples_per_chunk_runs)) + b"".join(struct.pack(">III", fc, spc, 1) for fc, spc in samples_per_chunk_runs))
    stsz = _fullbox(b"stsz", 0, 0, struct.pack(">II", 0, T) + b"".join(struct.pack(">I", z) for z in sizes))
    co64 = _fullbox(b"co64", 0, 0, struct.pack(">I", len(chunk_offsets)) + b"".join(struct.pack(">Q", o) for o in chunk_offsets))
    mhd = _fullbox(b"vmhd", 0, 1, bytes(8)) if handler == b"vide" else _fullbox(b"smhd", 0, 0, bytes(4))
    dinf = _box(b"dinf", _fullbox(b"dref", 0, 0, struct.pack(">I", 1) + _fullbox(b"url ", 0, 1, b"")))
    minf = _box(b"minf", mhd + dinf + _box(b"stbl", stsd + stts + stsc + stsz + co64))
    return _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))


def test_mp4_assembled_from_the_iso_bmff_spec(tmp_path):
    fr = _frames(T=5, H=12, W=16, seed=3)
    T, H, W, _ = fr.shape
    samples = [_png(f) for f in fr]
    audio = [bytes(20), bytes(21)]
    # VisualSampleEntry ('png '): 6 reserved, data_reference_index, 16 pre-defined / reserved, width, height, 72 dpi twice, reserved, frame_count,
    # 32-byte compressor name, depth 24, pre_defined -1
    vse = bytes(6) + struct.pack(">H", 1) + bytes(16) + struct.pack(">HH", W, H) + struct.pack(">II", 0x00480000, 0x00480000) + bytes(4) + struct.pack(">H", 1) + \
        bytes(32) + struct.pack(">Hh", 24, -1)
    vid_entry = struct.pack(">I4s", 8 + len(vse), b"png ") + vse
    ase = bytes(6) + struct.pack(">H", 1) + bytes(8) + struct.pack(">HHHH", 1, 8, 0, 0) + struct.pack(">I", 8000 << 16)
    aud_entry = struct.pack(">I4s", 8 + len(ase), b"raw ") + ase
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2")
    free = _box(b"free", bytes(11))
    # layout: ftyp | moov | free | mdat (64-bit header): [video 0, video 1 | audio 0 | video 2, video 3 | audio 1 | video 4] -> video chunks of 2, 2, 1 samples

    def build(moov_len):
        data0 = len(ftyp) + moov_len + len(free) + 16
        off, voffs, aoffs, order = data0, [], [], [("v", [0, 1]), ("a", [0]), ("v", [2, 3]), ("a", [1]), ("v", [4])]
        for kind, ids in order:
            (voffs if kind == "v" else aoffs).append(off)
            off += sum(len(samples[i]) if kind == "v" else len(audio[i]) for i in ids)
        vtrak = _trak(b"vide", vid_entry, [(1, 2), (3, 1)], [len(x) for x in samples], voffs, 30000, [(3, 1001), (2, 2002)], W, H, 2)
        atrak = _trak(b"soun", aud_entry, [(1, 1)], [len(x) for x in audio], aoffs, 8000, [(2, 20)], 0, 0, 1)
        mvhd = _fullbox(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 1000, 234) + struct.pack(">iH", 0x10000, 0x100) + bytes(10) + _UNITY + bytes(24) + struct.pack(">I", 3))
        moov = _box(b"moov", mvhd + atrak + vtrak)                     # the audio track first: the reader has to find the 'vide' handler
        mdat = _box(b"mdat", b"".join(samples[i] if kind == "v" else audio[i] for kind, ids in order for i in ids), large=True)
        return moov, mdat

    moov, _ = build(0)
    moov, mdat = build(len(moov))                                      # (chunk offsets depend on the length of 'moov', which does not depend on them)
    p = str(tmp_path / "spec.mp4")
    with open(p, "wb") as f:
        f.write(ftyp + moov + free + mdat)
    r = V.open_container(p)
    assert isinstance(r, V.Mp4Reader) and len(r) == T and (r.height, r.width) == (H, W)
    assert np.array_equal(r.get_batch([4, 0, 2]).asnumpy(), fr[[4, 0, 2]])
    assert np.array_equal(r.get_batch(list(range(T))).asnumpy(), fr)
    # 5 samples over 3 x 1001 + 2 x 2002 ticks of a 30000 Hz time scale
    assert abs(r.get_avg_fps() - 5 / (7007 / 30000)) < 1e-6


def test_mp4_h264_track_goes_to_decord(tmp_path):
    H, W = 16, 16
    vse = bytes(6) + struct.pack(">H", 1) + bytes(16) + struct.pack(">HH", W, H) + struct.pack(">II", 0x00480000, 0x00480000) + bytes(4) + struct.pack(">H", 1) + \
        bytes(32) + struct.pack(">Hh", 24, -1)
    entry = struct.pack(">I4s", 8 + len(vse), b"avc1") + vse
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomavc1")
    trak = _trak(b"vide", entry, [(1, 1)], [10], [len(ftyp) + 8], 25, [(1, 1)], W, H, 1)
    mvhd = _fullbox(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 25, 1) + struct.pack(">iH", 0x10000, 0x100) + bytes(10) + _UNITY + bytes(24) + struct.pack(">I", 2))
    p = str(tmp_path / "h264.mp4")
    with open(p, "wb") as f:
        f.write(ftyp + _box(b"mdat", bytes(10)) + _box(b"moov", mvhd + trak))
    with pytest.raises(V.NeedsDecoder, match="avc1"):
        V.Mp4Reader(p)
    try:
        import decord  # noqa: F401
        assert V.open_container(p) is None
    except ImportError:
        with pytest.raises(ImportError, match="avc1"):
            V.open_container(p)


def test_truncated_mp4_is_handed_on_not_raised(tmp_path):
    """a file these parsers cannot make sense of (here: cut inside 'moov') must reach decord; only without decord does an error surface, as ImportError"""
    fr = _frames(T=2, H=8, W=8)
    p = str(tmp_path / "ok.mp4")
    V.write_mjpeg_mp4(p, fr)
    raw = open(p, "rb").read()
    q = str(tmp_path / "cut.mp4")
    with open(q, "wb") as f:
        f.write(raw[: len(raw) - 40])
    try:
        import decord  # noqa: F401
        assert V.open_container(q) is None
    except ImportError:
        with pytest.raises(ImportError, match="not readable here"):
            V.open_container(q)


# ------------------------------------------------------------------------------------------------------------ YUV4MPEG2 with the optional header tags
def test_y4m_with_interlace_aspect_and_comment_tags(tmp_path):
    """the stream header of the y4m format allows I (interlacing), A (pixel aspect) and X (comment) tags in any order, and FRAME lines may carry
    parameters; write_y4m emits one fixed form.  4:4:4, BT.601 limited range, one grey and one saturated frame built by hand."""
    H, W = 4, 6
    planes = []
    for y_val, u_val, v_val in ((126, 128, 128), (81, 90, 240)):        # mid grey; the BT.601 code for pure red (R = 255)
        planes.append(bytes([y_val]) * (H * W) + bytes([u_val]) * (H * W) + bytes([v_val]) * (H * W))
    p = str(tmp_path / "tags.y4m")
    with open(p, "wb") as f:
        f.write(b"YUV4MPEG2 C444 XYSCSS=444 W6 H4 A1:1 It F30000:1001 XCOLORRANGE=LIMITED\n")
        f.write(b"FRAME\n" + planes[0])
        f.write(b"FRAME Ip\n" + planes[1])
    r = V.open_container(p)
    assert len(r) == 2 and (r.height, r.width) == (H, W) and abs(r.get_avg_fps() - 30000 / 1001) < 1e-9
    a = r.get_batch([0, 1]).asnumpy()
    assert np.abs(a[0].astype(int) - 128).max() <= 2                                  # Y = 126 -> (126 - 16) * 255 / 219 = 128
    assert a[1][..., 0].min() >= 250 and a[1][..., 1].max() <= 5 and a[1][..., 2].max() <= 5

"""Container readers for `process_video` that need no decord / imageio / moviepy (trace/mm_utils.py:379-449 opens `.mp4 / .gif / .webm`
through those packages; none of them exists in this image, and a codec such as H.264 is out of this build's scope).  What is read here:

* `.y4m` (YUV4MPEG2, the uncompressed interchange format `ffmpeg -i clip.mp4 clip.y4m` writes): 4:2:0 / 4:2:2 / 4:4:4 / mono, 8 bit;
* `.avi` holding Motion-JPEG (`MJPG`) or uncompressed 24-bit (`DIB `) frames: every frame is an independent image, decoded with Pillow;
* `.mp4 / .mov / .m4v` (ISO-BMFF) whose video track holds still images (Motion-JPEG / Photo-JPEG / PNG samples): the sample table is parsed here,
  the samples decoded with Pillow; an inter-coded track (avc1, hvc1, vp09, ...) is named and handed to decord, as the reference does.

Both readers present decord's `VideoReader` surface as far as `process_video` uses it — `len(vr)`, `vr.get_avg_fps()`,
`vr.get_batch(indices).asnumpy()` -> uint8 `[n, H, W, 3]` RGB (mm_utils.py:421-431) — so the sampling / timestamp code above them is the
same code that runs over decord.  YUV -> RGB is BT.601 limited range (what libswscale assumes for untagged SD material), integer
arithmetic, chroma replicated to full resolution (nearest); `rgb_to_yuv601` / `yuv601_to_rgb` state the exact formulas and
`tests/test_video_io.py` pins them.  A writer for each format serves the tests and `bench.py --config c1` (a synthetic clip on disk).
"""
from __future__ import annotations

import io
import os
import struct
from typing import List, Sequence, Tuple

import numpy as np


class _Batch:
    """what decord's get_batch returns, reduced to the one method the callers use"""

    def __init__(self, arr: np.ndarray):
        self._a = arr

    def asnumpy(self) -> np.ndarray:
        return self._a

    def numpy(self) -> np.ndarray:
        return self._a


# ---------------------------------------------------------------------------------------------------------------- colour
def yuv601_to_rgb(y: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """BT.601 limited range, 8 bit, integer arithmetic (16.16 fixed point, rounded):  C = Y - 16, D = U - 128, E = V - 128,
    R = 1.164383 C + 1.596027 E,  G = 1.164383 C - 0.391762 D - 0.812968 E,  B = 1.164383 C + 2.017232 D;  planes of equal shape."""
    c = y.astype(np.int64) - 16
    d = u.astype(np.int64) - 128
    e = v.astype(np.int64) - 128
    r = (76309 * c + 104597 * e + 32768) >> 16
    g = (76309 * c - 25675 * d - 53279 * e + 32768) >> 16
    b = (76309 * c + 132201 * d + 32768) >> 16
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def rgb_to_yuv601(rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """inverse convention (writer side): Y = 16 + 0.256788 R + 0.504129 G + 0.097906 B, U = 128 - 0.148223 R - 0.290993 G + 0.439216 B,
    V = 128 + 0.439216 R - 0.367788 G - 0.071427 B; 16.16 fixed point, rounded."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    y = 16 + ((16829 * r + 33039 * g + 6416 * b + 32768) >> 16)
    u = 128 + ((-9714 * r - 19070 * g + 28784 * b + 32768) >> 16)
    v = 128 + ((28784 * r - 24103 * g - 4681 * b + 32768) >> 16)
    return tuple(np.clip(p, 0, 255).astype(np.uint8) for p in (y, u, v))


# ---------------------------------------------------------------------------------------------------------------- YUV4MPEG2
_Y4M_SUB = {"420": (2, 2), "422": (2, 1), "444": (1, 1), "mono": (0, 0)}          # chroma subsampling (horizontal, vertical)


class Y4MReader:
    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            head = f.readline(4096)
            if not head.startswith(b"YUV4MPEG2 ") or not head.endswith(b"\n"):
                raise ValueError(f"{path}: not a YUV4MPEG2 stream")
            self._data0 = f.tell()
        w = h = 0
        fps = (25, 1)
        chroma = "420"
        for tok in head[len(b"YUV4MPEG2 "):].split():
            tag, val = chr(tok[0]), tok[1:].decode("ascii", "replace")
            if tag == "W":
                w = int(val)
            elif tag == "H":
                h = int(val)
            elif tag == "F":
                n, _, d = val.partition(":")
                fps = (int(n), int(d or 1))
            elif tag == "C":
                chroma = val
        key = {"420": "420", "420jpeg": "420", "420mpeg2": "420", "420paldv": "420", "422": "422", "444": "444", "mono": "mono"}.get(chroma)
        if key is None:                                    # (420p10, 444p16, 444alpha, ...: not 8-bit three-plane)
            raise ValueError(f"{path}: unsupported y4m colour space C{chroma} (8-bit 420 / 422 / 444 / mono only)")
        if w < 1 or h < 1 or fps[0] < 1 or fps[1] < 1:
            raise ValueError(f"{path}: bad y4m header {head!r}")
        self.width, self.height, self._fps, self._key = w, h, fps[0] / fps[1], key
        sx, sy = _Y4M_SUB[key]
        self._cw = 0 if not sx else (w + sx - 1) // sx
        self._ch = 0 if not sy else (h + sy - 1) // sy
        self._frame_bytes = w * h + 2 * self._cw * self._ch
        # frame offsets: every frame is "FRAME[ params]\n" + planes; headers may carry parameters, so the file is walked once
        self._offsets: List[int] = []
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            pos = self._data0
            while pos < size:
                f.seek(pos)
                line = f.readline(256)
                if not line.startswith(b"FRAME") or not line.endswith(b"\n"):
                    raise ValueError(f"{path}: malformed frame header at byte {pos}")
                pos += len(line)
                if pos + self._frame_bytes > size:
                    break                                  # truncated last frame: dropped, as decoders do
                self._offsets.append(pos)
                pos += self._frame_bytes

    def __len__(self) -> int:
        return len(self._offsets)

    def get_avg_fps(self) -> float:
        return self._fps

    def _frame(self, f, i: int) -> np.ndarray:
        f.seek(self._offsets[i])
        buf = np.frombuffer(f.read(self._frame_bytes), dtype=np.uint8)
        w, h, cw, ch = self.width, self.height, self._cw, self._ch
        y = buf[: w * h].reshape(h, w)
        if self._key == "mono":
            u = v = np.full((h, w), 128, np.uint8)
        else:
            sx, sy = _Y4M_SUB[self._key]
            u = buf[w * h: w * h + cw * ch].reshape(ch, cw).repeat(sy, axis=0).repeat(sx, axis=1)[:h, :w]
            v = buf[w * h + cw * ch:].reshape(ch, cw).repeat(sy, axis=0).repeat(sx, axis=1)[:h, :w]
        return yuv601_to_rgb(y, u, v)

    def get_batch(self, indices: Sequence[int]) -> _Batch:
        idx = [int(i) for i in indices]
        if any(i < 0 or i >= len(self) for i in idx):
            raise IndexError(f"frame index out of range (0..{len(self) - 1}): {idx}")
        with open(self.path, "rb") as f:
            return _Batch(np.stack([self._frame(f, i) for i in idx]) if idx else np.zeros((0, self.height, self.width, 3), np.uint8))

    def __getitem__(self, i: int) -> np.ndarray:
        return self.get_batch([i]).asnumpy()[0]


def write_y4m(path: str, frames_rgb: np.ndarray, fps: Tuple[int, int] = (25, 1), chroma: str = "420") -> None:
    """uint8 RGB [T, H, W, 3] -> YUV4MPEG2 (BT.601 limited range; 4:2:0 / 4:2:2 chroma = the mean of the covered pixels, rounded)"""
    fr = np.asarray(frames_rgb)
    if fr.dtype != np.uint8 or fr.ndim != 4 or fr.shape[-1] != 3:
        raise ValueError("frames must be uint8 [T, H, W, 3]")
    sx, sy = _Y4M_SUB[chroma]
    T, H, W, _ = fr.shape
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F{fps[0]}:{fps[1]} Ip A1:1 C{chroma}{'jpeg' if chroma == '420' else ''}\n".encode())
        for t in range(T):
            y, u, v = rgb_to_yuv601(fr[t])
            f.write(b"FRAME\n")
            f.write(y.tobytes())
            if chroma == "mono":
                continue
            for p in (u, v):
                if sx > 1 or sy > 1:
                    Hp, Wp = (H + sy - 1) // sy * sy, (W + sx - 1) // sx * sx
                    q = np.pad(p, ((0, Hp - H), (0, Wp - W)), mode="edge").astype(np.int64)
                    q = q.reshape(Hp // sy, sy, Wp // sx, sx).sum(axis=(1, 3))
                    p = ((q + (sx * sy) // 2) // (sx * sy)).astype(np.uint8)
                f.write(p.tobytes())


class NeedsDecoder(ValueError):
    """The container is well-formed but its video stream is inter-coded (H.264, MPEG-4, VP9, ...): not something these readers decode.
    open_container() hands such a file to decord, exactly as the reference does for every file (mm_utils.py:421)."""


# ---------------------------------------------------------------------------------------------------------------- AVI (MJPG / DIB)
def _riff_chunks(buf: memoryview, start: int, end: int):
    """(fourcc, payload start, payload size) of every chunk in [start, end); LIST chunks are yielded with their list type appended"""
    pos = start
    while pos + 8 <= end:
        cc = bytes(buf[pos:pos + 4])
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        yield cc, pos + 8, size
        pos += 8 + size + (size & 1)


class AviReader:
    """AVI (1.0 and OpenDML: further 'RIFF....AVIX' segments, 'rec ' groups inside 'movi') with one video stream of independent frames: Motion-JPEG
    ('MJPG' and its aliases) or uncompressed 24-bit BGR ('DIB ', BI_RGB, bottom-up).  A zero-length video chunk is a dropped frame: it shows the
    previous picture again and keeps its place on the time line.  Inter-coded streams (H.264, MPEG-4, ...) raise NeedsDecoder."""

    def __init__(self, path: str):
        import mmap
        self.path = path
        self._file = open(path, "rb")
        try:                                               # mapped, not read: a Motion-JPEG clip can be gigabytes, and only the sampled frames are touched
            data = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:                                 # (an empty file cannot be mapped)
            self._file.close()
            raise ValueError(f"{path}: not a RIFF AVI file")
        buf = memoryview(data)
        if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"AVI ":
            buf.release(); data.close(); self._file.close()
            raise ValueError(f"{path}: not a RIFF AVI file")
        self._data = data
        try:
            self._frames: List[Tuple[int, int]] = []
            self._fps = 0.0
            self._handler = b""
            self._compression = b""
            self.width = self.height = 0
            self._bits = 24
            usec = 0

            def movi(start: int, end: int) -> None:
                for c2, p2, n2 in _riff_chunks(buf, start, end):
                    if c2 == b"LIST" and bytes(buf[p2:p2 + 4]) == b"rec ":
                        movi(p2 + 4, min(p2 + n2, end))                # OpenDML interleave groups
                    elif c2[2:] in (b"dc", b"db"):
                        if n2 > 0:
                            self._frames.append((p2, n2))
                        elif self._frames:
                            self._frames.append(self._frames[-1])      # dropped frame: the previous picture stays up
            # the first RIFF chunk ('AVI ') holds the headers and the first 'movi' list; OpenDML files continue in 'RIFF....AVIX' chunks of further 'movi' lists
            segs, pos = [], 0
            while pos + 12 <= len(data) and data[pos:pos + 4] == b"RIFF":
                size = struct.unpack_from("<I", buf, pos + 4)[0]
                segs.append((bytes(buf[pos + 8:pos + 12]), pos + 12, min(pos + 8 + size, len(data))))
                pos += 8 + size + (size & 1)
            for form, s0, s1 in segs[1:]:
                if form != b"AVIX":
                    segs = segs[:1]
                    break
            for cc, p, n in (x for form, s0, s1 in segs for x in _riff_chunks(buf, s0, s1)):
                if cc != b"LIST":
                    continue
                kind = bytes(buf[p:p + 4])
                if kind == b"hdrl":
                    for c2, p2, n2 in _riff_chunks(buf, p + 4, p + n):
                        if c2 == b"avih":
                            usec = struct.unpack_from("<I", buf, p2)[0]
                        elif c2 == b"LIST" and bytes(buf[p2:p2 + 4]) == b"strl":
                            is_video = False
                            for c3, p3, n3 in _riff_chunks(buf, p2 + 4, p2 + n2):
                                if c3 == b"strh":
                                    is_video = bytes(buf[p3:p3 + 4]) == b"vids"
                                    if is_video and not self._handler:
                                        self._handler = bytes(buf[p3 + 4:p3 + 8])
                                        scale, rate = struct.unpack_from("<II", buf, p3 + 20)
                                        if scale and rate:
                                            self._fps = rate / scale
                                elif c3 == b"strf" and is_video and not self.width:
                                    self.width, h = struct.unpack_from("<ii", buf, p3 + 4)
                                    self.height = abs(h)
                                    self._flip = h > 0                 # positive height = bottom-up rows (DIB)
                                    self._bits = struct.unpack_from("<H", buf, p3 + 14)[0]
                                    self._compression = bytes(buf[p3 + 16:p3 + 20])
                elif kind == b"movi":
                    movi(p + 4, min(p + n, len(data)))
        except Exception:                                  # a parse error must not leave the map, the view or the descriptor behind (the caller falls back to decord)
            buf.release(); self.close()
            raise
        buf.release()
        if not self._fps and usec:
            self._fps = 1e6 / usec
        comp = self._compression.upper()
        self._mjpeg = comp in (b"MJPG", b"JPEG", b"AVRN", b"LJPG") or self._handler.upper() in (b"MJPG",)
        self._dib = comp in (b"\0\0\0\0", b"DIB ", b"RGB ", b"RAW ") and self._bits == 24
        if not (self._mjpeg or self._dib):
            self.close()
            raise NeedsDecoder(f"{path}: video stream is coded as {self._compression!r} / {self._handler!r}: only streams of independent frames (Motion-JPEG, "
                             "uncompressed 24-bit) are decoded here; transcode inter-coded video (H.264, ...) with `ffmpeg -i in.mp4 out.y4m`, or install decord")
        if not self._frames or self.width < 1 or self.height < 1 or self._fps <= 0:
            self.close()
            raise ValueError(f"{path}: no video frames / bad stream header")

    def __len__(self) -> int:
        return len(self._frames)

    def close(self) -> None:
        if getattr(self, "_data", None) is not None:
            self._data.close(); self._file.close()
            self._data = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_avg_fps(self) -> float:
        return self._fps

    def _frame(self, i: int) -> np.ndarray:
        p, n = self._frames[i]
        raw = self._data[p:p + n]
        if self._mjpeg:
            from PIL import Image
            with Image.open(io.BytesIO(raw)) as im:
                return np.asarray(im.convert("RGB"))
        stride = (self.width * 3 + 3) & ~3
        a = np.frombuffer(raw, dtype=np.uint8)[: stride * self.height].reshape(self.height, stride)[:, : self.width * 3]
        a = a.reshape(self.height, self.width, 3)[:, :, ::-1]
        return np.ascontiguousarray(a[::-1] if self._flip else a)

    def get_batch(self, indices: Sequence[int]) -> _Batch:
        idx = [int(i) for i in indices]
        if any(i < 0 or i >= len(self) for i in idx):
            raise IndexError(f"frame index out of range (0..{len(self) - 1}): {idx}")
        return _Batch(np.stack([self._frame(i) for i in idx]) if idx else np.zeros((0, self.height, self.width, 3), np.uint8))

    def __getitem__(self, i: int) -> np.ndarray:
        return self._frame(int(i))


def write_avi(path: str, frames_rgb: np.ndarray, fps: Tuple[int, int] = (25, 1), codec: str = "MJPG", quality: int = 95) -> None:
    """uint8 RGB [T, H, W, 3] -> AVI with Motion-JPEG (Pillow's encoder) or uncompressed bottom-up 24-bit BGR frames ('DIB ')"""
    fr = np.asarray(frames_rgb)
    if fr.dtype != np.uint8 or fr.ndim != 4 or fr.shape[-1] != 3:
        raise ValueError("frames must be uint8 [T, H, W, 3]")
    T, H, W, _ = fr.shape
    chunks = []
    for t in range(T):
        if codec == "MJPG":
            from PIL import Image
            b = io.BytesIO()
            Image.fromarray(fr[t]).save(b, format="JPEG", quality=quality, subsampling=0)
            payload = b.getvalue()
        else:
            stride = (W * 3 + 3) & ~3
            rows = np.zeros((H, stride), np.uint8)
            rows[:, : W * 3] = fr[t][::-1, :, ::-1].reshape(H, W * 3)
            payload = rows.tobytes()
        chunks.append(payload)

    def chunk(cc: bytes, payload: bytes) -> bytes:
        return cc + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")

    def lst(kind: bytes, body: bytes) -> bytes:
        return b"LIST" + struct.pack("<I", 4 + len(body)) + kind + body

    comp = b"MJPG" if codec == "MJPG" else b"\0\0\0\0"
    handler = b"MJPG" if codec == "MJPG" else b"DIB "
    avih = struct.pack("<14I", int(round(1e6 * fps[1] / fps[0])), 0, 0, 0x10, T, 0, 1, max(len(c) for c in chunks), W, H, 0, 0, 0, 0)
    strh = b"vids" + handler + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, fps[1], fps[0], 0, T, max(len(c) for c in chunks), 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, W, H)
    strf = struct.pack("<IiiHH", 40, W, H, 1, 24) + comp + struct.pack("<IiiII", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi = lst(b"movi", b"".join(chunk(b"00dc" if codec == "MJPG" else b"00db", c) for c in chunks))
    body = b"AVI " + hdrl + movi
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


# ---------------------------------------------------------------------------------------------------------------- MP4 / MOV (intra-only codecs)
_MP4_INTRA = {b"jpeg", b"mjpa", b"mjpg", b"MJPG", b"png "}         # sample = one still image Pillow decodes (Photo-JPEG / Motion-JPEG A / PNG)


def _boxes(buf, start: int, end: int):
    """(type, payload start, payload end) of the ISO-BMFF boxes in [start, end)"""
    pos = start
    while pos + 8 <= end:
        size, typ = struct.unpack_from(">I4s", buf, pos)
        hdr = 8
        if size == 1:
            size = struct.unpack_from(">Q", buf, pos + 8)[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr or pos + size > end:
            break
        yield typ, pos + hdr, pos + size
        pos += size


class Mp4Reader:
    """ISO-BMFF (`.mp4 / .mov / .m4v`) with a video track of independent still images (Motion-JPEG / Photo-JPEG / PNG samples): sample table -> byte
    ranges -> Pillow.  A track coded with an inter-frame codec (avc1 = H.264, hvc1 = HEVC, vp09, av01, mp4v ...) is refused with its fourcc: that
    needs a video decoder (decord in the reference), not a container parser."""

    def __init__(self, path: str):
        import mmap
        self.path = path
        self._file = open(path, "rb")
        try:
            data = mmap.mmap(self._file.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._file.close()
            raise ValueError(f"{path}: not an ISO base media file")
        self._data = data
        buf = memoryview(data)
        try:
            self._parse(buf, path)
        except Exception:
            buf.release(); self.close()
            raise
        buf.release()

    def _parse(self, buf, path):
        n = len(buf)
        top = {t: (a, b) for t, a, b in _boxes(buf, 0, n)}
        if b"moov" not in top:
            raise ValueError(f"{path}: not an ISO base media file (no moov box)")
        track = None
        for t, a, b in _boxes(buf, *top[b"moov"]):
            if t != b"trak":
                continue
            mdia = next(((x, y) for tt, x, y in _boxes(buf, a, b) if tt == b"mdia"), None)
            if mdia is None:
                continue
            sub = {tt: (x, y) for tt, x, y in _boxes(buf, *mdia)}
            if b"hdlr" not in sub or bytes(buf[sub[b"hdlr"][0] + 8: sub[b"hdlr"][0] + 12]) != b"vide" or b"minf" not in sub:
                continue
            minf = {tt: (x, y) for tt, x, y in _boxes(buf, *sub[b"minf"])}
            if b"stbl" not in minf:
                continue
            track = (sub, {tt: (x, y) for tt, x, y in _boxes(buf, *minf[b"stbl"])})
            break
        if track is None:
            raise ValueError(f"{path}: no video track")
        sub, stbl = track
        m0 = sub[b"mdhd"][0]
        ver = buf[m0]
        timescale, duration = (struct.unpack_from(">IQ", buf, m0 + 20) if ver == 1 else struct.unpack_from(">II", buf, m0 + 12))
        s0 = stbl[b"stsd"][0]
        fourcc = bytes(buf[s0 + 12: s0 + 16])
        self.width, self.height = struct.unpack_from(">HH", buf, s0 + 8 + 8 + 24)
        if fourcc not in _MP4_INTRA:
            raise NeedsDecoder(f"{path}: the video track is coded as {fourcc!r}: only tracks of independent still images (Motion-JPEG / Photo-JPEG / PNG samples) "
                             "are decoded here; inter-coded video (avc1 = H.264, hvc1, vp09, av01, ...) needs a decoder — `ffmpeg -i in.mp4 out.y4m`, or install decord")
        z0 = stbl[b"stsz"][0]
        uniform, count = struct.unpack_from(">II", buf, z0 + 4)
        sizes = [uniform] * count if uniform else list(struct.unpack_from(f">{count}I", buf, z0 + 12))
        if b"stco" in stbl:
            c0 = stbl[b"stco"][0]
            nch = struct.unpack_from(">I", buf, c0 + 4)[0]
            chunks = list(struct.unpack_from(f">{nch}I", buf, c0 + 8))
        else:
            c0 = stbl[b"co64"][0]
            nch = struct.unpack_from(">I", buf, c0 + 4)[0]
            chunks = list(struct.unpack_from(f">{nch}Q", buf, c0 + 8))
        c0 = stbl[b"stsc"][0]
        nsc = struct.unpack_from(">I", buf, c0 + 4)[0]
        runs = [struct.unpack_from(">III", buf, c0 + 8 + 12 * i) for i in range(nsc)]          # (first chunk, samples per chunk, description)
        self._frames: List[Tuple[int, int]] = []
        si = 0
        for ci in range(nch):
            per = next(r[1] for r in reversed(runs) if r[0] <= ci + 1)
            off = chunks[ci]
            for _ in range(per):
                if si >= count:
                    break
                self._frames.append((off, sizes[si]))
                off += sizes[si]
                si += 1
        if si != count or not count:
            raise ValueError(f"{path}: sample table does not add up ({si} of {count} samples placed)")
        # average frame rate = samples / (sum of the stts deltas / timescale); the media duration as the fallback
        t0 = stbl[b"stts"][0]
        nts = struct.unpack_from(">I", buf, t0 + 4)[0]
        total = sum(c * d for c, d in (struct.unpack_from(">II", buf, t0 + 8 + 8 * i) for i in range(nts))) or duration
        if not total or not timescale:
            raise ValueError(f"{path}: no timing information")
        self._fps = count * timescale / total

    def __len__(self) -> int:
        return len(self._frames)

    def get_avg_fps(self) -> float:
        return self._fps

    def close(self) -> None:
        if getattr(self, "_data", None) is not None:
            self._data.close(); self._file.close()
            self._data = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getitem__(self, i: int) -> np.ndarray:
        from PIL import Image
        p, n = self._frames[int(i)]
        with Image.open(io.BytesIO(self._data[p:p + n])) as im:
            return np.asarray(im.convert("RGB"))

    def get_batch(self, indices: Sequence[int]) -> _Batch:
        idx = [int(i) for i in indices]
        if any(i < 0 or i >= len(self) for i in idx):
            raise IndexError(f"frame index out of range (0..{len(self) - 1}): {idx}")
        return _Batch(np.stack([self[i] for i in idx]) if idx else np.zeros((0, self.height, self.width, 3), np.uint8))


def write_mjpeg_mp4(path: str, frames_rgb: np.ndarray, fps: Tuple[int, int] = (25, 1), quality: int = 95, fourcc: bytes = b"jpeg") -> None:
    """uint8 RGB [T, H, W, 3] -> a minimal ISO-BMFF file with one Photo-JPEG video track (one chunk per sample) — for the tests and for round trips"""
    from PIL import Image
    fr = np.asarray(frames_rgb)
    if fr.dtype != np.uint8 or fr.ndim != 4 or fr.shape[-1] != 3:
        raise ValueError("frames must be uint8 [T, H, W, 3]")
    T, H, W, _ = fr.shape
    samples = []
    for t in range(T):
        b = io.BytesIO()
        Image.fromarray(fr[t]).save(b, format="JPEG", quality=quality, subsampling=0)
        samples.append(b.getvalue())

    def box(typ: bytes, payload: bytes) -> bytes:
        return struct.pack(">I4s", 8 + len(payload), typ) + payload

    def full(typ: bytes, payload: bytes, version: int = 0, flags: int = 0) -> bytes:
        return box(typ, struct.pack(">I", (version << 24) | flags) + payload)

    ftyp = box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    mdat_payload = b"".join(samples)
    data0 = len(ftyp) + 8
    offs, o = [], data0
    for sm in samples:
        offs.append(o)
        o += len(sm)
    timescale, delta = fps[0], fps[1]
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = full(b"mvhd", struct.pack(">IIII", 0, 0, timescale, T * delta) + struct.pack(">IH", 0x10000, 0x100) + bytes(10) + matrix + bytes(24) + struct.pack(">I", 2))
    tkhd = full(b"tkhd", struct.pack(">IIIII", 0, 0, 1, 0, T * delta) + bytes(8) + struct.pack(">HHHH", 0, 0, 0, 0) + matrix + struct.pack(">II", W << 16, H << 16), flags=3)
    mdhd = full(b"mdhd", struct.pack(">IIII", 0, 0, timescale, T * delta) + struct.pack(">HH", 0x55C4, 0))
    hdlr = full(b"hdlr", struct.pack(">I4s", 0, b"vide") + bytes(12) + b"VideoHandler\0")
    entry = bytes(6) + struct.pack(">H", 1) + bytes(16) + struct.pack(">HH", W, H) + struct.pack(">II", 0x480000, 0x480000) + struct.pack(">I", 0) + \
        struct.pack(">H", 1) + bytes(32) + struct.pack(">Hh", 24, -1)
    stsd = full(b"stsd", struct.pack(">I", 1) + struct.pack(">I4s", 8 + len(entry), fourcc) + entry)
    stts = full(b"stts", struct.pack(">III", 1, T, delta))
    stsc = full(b"stsc", struct.pack(">IIII", 1, 1, 1, 1))
    stsz = full(b"stsz", struct.pack(">II", 0, T) + struct.pack(f">{T}I", *[len(x) for x in samples]))
    stco = full(b"stco", struct.pack(">I", T) + struct.pack(f">{T}I", *offs))
    stbl = box(b"stbl", stsd + stts + stsc + stsz + stco)
    dinf = box(b"dinf", full(b"dref", struct.pack(">I", 1) + full(b"url ", b"", flags=1)))
    minf = box(b"minf", full(b"vmhd", bytes(8), flags=1) + dinf + stbl)
    moov = box(b"moov", mvhd + box(b"trak", tkhd + box(b"mdia", mdhd + hdlr + minf)))
    with open(path, "wb") as f:
        f.write(ftyp + box(b"mdat", mdat_payload) + moov)


def open_container(path: str):
    """A reader with decord's VideoReader surface for the containers decodable without a codec library, or None (the caller then goes to decord, as the
    reference does for every file: mm_utils.py:421).  The readers here are a FAST PATH, never a gate: an .avi / .mp4 / .mov whose stream is inter-coded
    (NeedsDecoder) or that these parsers cannot make sense of (a truncated or unusual box / chunk structure: ValueError, KeyError, IndexError, struct.error)
    is handed on; only when decord is absent does the reader's own message surface, as an ImportError that names the codec or the parse problem."""
    low = path.lower()
    if low.endswith(".y4m"):
        return Y4MReader(path)
    reader = AviReader if low.endswith(".avi") else Mp4Reader if low.endswith((".mp4", ".mov", ".m4v")) else None
    if reader is None:
        return None
    try:
        return reader(path)
    except (ValueError, KeyError, IndexError, struct.error) as e:          # NeedsDecoder is a ValueError
        try:
            import decord  # noqa: F401
        except ImportError:
            raise ImportError(f"{e}" if isinstance(e, NeedsDecoder) else f"{path}: not readable here ({type(e).__name__}: {e}) and decord is not installed") from e
        return None

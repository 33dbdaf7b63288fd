"""The step after the trace loop in the reference's examples (examples/scene.rs:111-119): the accumulated
radiance becomes an 8-bit image file, and - not in the reference, whose examples render in one go - a render can be
stopped and resumed by sample range.

* `write_png` / `read_png`: RGB8 PNG with zlib from the standard library (the examples call `image::save("test.jpg")`;
  the JPEG codec is a third-party crate and stays out of scope, PNG keeps the file lossless and testable).
* `save_accum` / `load_accum`: the f32 accumulation image plus what identifies the render (size, depth, seed,
  samples done) as an .npz without pickled objects.
* `render_progressive`: renders samples [done, spp_total) in chunks, saving after each; a later call with the same
  file continues where it stopped.  Sample k always draws the Philox stream keyed by k, so the result equals a
  one-shot render up to the f32 order of the per-chunk sums.
"""
import os
import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, rgb8):
    """rgb8: (H, W, 3) uint8, row 0 at the top (the examples' `raw[h * WIDTH + w]` order)."""
    a = np.ascontiguousarray(rgb8)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_png wants an (H, W, 3) uint8 array")
    h, w, _ = a.shape
    rows = np.empty((h, 1 + 3 * w), np.uint8)
    rows[:, 0] = 0                      # filter type 0 (None) on every scanline
    rows[:, 1:] = a.reshape(h, 3 * w)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)   # 8 bit, colour type 2 (RGB)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + _chunk(b"IEND", b""))


def read_png(path):
    """Reads back what `write_png` writes (RGB8, non-interlaced, filter 0-4); checks every CRC."""
    data = open(path, "rb").read()
    if data[:8] != _PNG_SIG:
        raise ValueError("not a PNG file")
    pos, idat, w, h = 8, b"", None, None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if crc != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError("PNG chunk %r fails its CRC" % tag)
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 2, 0):
                raise ValueError("only 8-bit RGB non-interlaced PNGs are read")
        elif tag == b"IDAT":
            idat += body
        elif tag == b"IEND":
            break
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), np.uint8)
    prev = np.zeros(3 * w, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:   # Sub, Average, Paeth need the already decoded pixel to the left
            cur = np.zeros(3 * w, np.int32)
            for x in range(3 * w):
                a = cur[x - 3] if x >= 3 else 0
                b = prev[x]
                c = prev[x - 3] if x >= 3 else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                elif ft == 4:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError("bad PNG filter type")
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)


def write_ppm(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def vec3_to_rgb_host(accum, divisor):
    """examples/scene.rs:122-127 on the host, in f64 like the reference: (255.99 * sqrt(clamp(c, 0, 1))) as u8."""
    c = np.clip(np.asarray(accum, np.float64) / float(divisor), 0.0, 1.0)
    return (255.99 * np.sqrt(c)).astype(np.uint8)


# ---------------------------------------------------------------------------------- resumable accumulation
_KEYS = ("width", "height", "depth", "seed", "spp_done")


def save_accum(path, accum, *, width, height, depth, seed, spp_done):
    """Atomic (write + rename) .npz: the f32 sum over samples [0, spp_done) and the render's identity."""
    accum = np.ascontiguousarray(accum, np.float32).reshape(height, width, 3)
    tmp = path + ".tmp.npz"
    np.savez(tmp, accum=accum, meta=np.array([width, height, depth, seed, spp_done], np.uint64))
    os.replace(tmp, path)


def load_accum(path):
    with np.load(path, allow_pickle=False) as z:
        meta = dict(zip(_KEYS, (int(v) for v in z["meta"])))
        accum = np.array(z["accum"], np.float32)
    if accum.shape != (meta["height"], meta["width"], 3):
        raise ValueError("accumulation file: image shape does not match its header")
    return accum, meta


def render_progressive(render_range, path, *, width, height, depth, seed, spp_total, chunk):
    """render_range(spp_begin, spp_end) -> (H, W, 3) f32 sum of those samples (e.g. a lambda around World.render).
    Continues from `path` if it exists (and belongs to the same render), saves after every chunk, returns
    (accum, spp_done)."""
    if chunk <= 0:
        raise ValueError("chunk must be positive")
    done, accum = 0, np.zeros((height, width, 3), np.float32)
    if os.path.exists(path):
        accum, meta = load_accum(path)
        want = dict(width=width, height=height, depth=depth, seed=seed)
        for k, v in want.items():
            if meta[k] != v:
                raise ValueError("accumulation file %s was made with %s=%d, not %d" % (path, k, meta[k], v))
        done = meta["spp_done"]
    while done < spp_total:
        end = min(done + chunk, spp_total)
        accum = accum + np.asarray(render_range(done, end), np.float32).reshape(height, width, 3)
        done = end
        save_accum(path, accum, width=width, height=height, depth=depth, seed=seed, spp_done=done)
    return accum, done

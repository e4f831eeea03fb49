"""Recorded-sensor reader and pool checkpoint header logic that need no GPU: PNG / PGM / PPM
decoding against images written here with Python's zlib (all five PNG filter types, stored / fixed / dynamic
deflate blocks), association-list parsing, depth unit conversion, focal from FOV (openni_device.cpp:64-65)."""
import math
import struct
import zlib

import numpy as np
import pytest



def load_pkg():
    import svoslam_pkg
    return svoslam_pkg.load()


def png_bytes(img, level=9, strategy=zlib.Z_DEFAULT_STRATEGY, filters=(0, 1, 2, 3, 4)):
    """minimal PNG writer: [h,w] uint16 grey or [h,w,3] uint8 RGB, filter type cycling over `filters`"""
    h, w = img.shape[:2]
    if img.dtype == np.uint16:
        ctype, depth, bpp = 0, 16, 2
        rows = img.astype(">u2").tobytes()
    else:
        ctype, depth, bpp = 2, 8, 3
        rows = img.astype(np.uint8).tobytes()
    stride = w * bpp
    raw = bytearray()
    prev = bytearray(stride)
    for y in range(h):
        cur = bytearray(rows[y * stride:(y + 1) * stride])
        ft = filters[y % len(filters)]
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pred) & 0xFF
        raw.append(ft)
        raw += out
        prev = cur
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    z = co.compress(bytes(raw)) + co.flush()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


@pytest.fixture(scope="module")
def images():
    rng = np.random.default_rng(7)
    h, w = 37, 53
    yy, xx = np.mgrid[0:h, 0:w]
    depth = (1500 + 40 * xx + 3 * yy * yy + rng.integers(0, 7, (h, w))).astype(np.uint16)
    depth[rng.random((h, w)) < 0.05] = 0
    rgb = np.stack([(xx * 5) % 256, (yy * 7 + xx) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    return depth, rgb


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (1, zlib.Z_HUFFMAN_ONLY), (9, zlib.Z_RLE)])
def test_png_decoding(tmp_path, images, level, strategy):
    pkg = load_pkg()
    depth, rgb = images
    (tmp_path / "d.png").write_bytes(png_bytes(depth, level, strategy))
    (tmp_path / "c.png").write_bytes(png_bytes(rgb, level, strategy))
    assert np.array_equal(pkg.image_load(tmp_path / "d.png"), depth)
    assert np.array_equal(pkg.image_load(tmp_path / "c.png"), rgb)


def test_png_noise_image_large(tmp_path):
    """incompressible data, > 32 KB (several deflate blocks, distances up to the window)"""
    pkg = load_pkg()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, (240, 320)).astype(np.uint16)
    img[100:140] = img[20:60]   # long-distance matches
    (tmp_path / "n.png").write_bytes(png_bytes(img, 9))
    assert np.array_equal(pkg.image_load(tmp_path / "n.png"), img)


def test_pnm_and_errors(tmp_path, images):
    pkg = load_pkg()
    depth, rgb = images
    h, w = depth.shape
    (tmp_path / "d.pgm").write_bytes(b"P5\n# comment\n%d %d\n65535\n" % (w, h) + depth.astype(">u2").tobytes())
    (tmp_path / "c.ppm").write_bytes(b"P6 %d %d 255\n" % (w, h) + rgb.tobytes())
    assert np.array_equal(pkg.image_load(tmp_path / "d.pgm"), depth)
    assert np.array_equal(pkg.image_load(tmp_path / "c.ppm"), rgb)
    bad = bytearray(png_bytes(depth))
    bad[60] ^= 0x55     # corrupt the deflate stream: Adler-32 / Huffman check must catch it
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    with pytest.raises(pkg.SvoslamError):
        pkg.image_load(tmp_path / "bad.png")
    with pytest.raises(pkg.SvoslamError):
        pkg.image_load(tmp_path / "missing.png")


def test_frame_reader_association_list(tmp_path, images):
    pkg = load_pkg()
    depth, rgb = images
    (tmp_path / "depth").mkdir(); (tmp_path / "rgb").mkdir()
    lines = ["# colour first, as associate.py rgb.txt depth.txt writes it"]
    frames = []
    for k in range(3):
        d = (depth + 11 * k).astype(np.uint16)
        c = np.roll(rgb, k, axis=1)
        (tmp_path / "depth" / ("%d.png" % k)).write_bytes(png_bytes(d))
        (tmp_path / "rgb" / ("%d.png" % k)).write_bytes(png_bytes(c, 6))
        lines.append("%.6f rgb/%d.png %.6f depth/%d.png" % (1305031102.175304 + k / 30, k, 1305031102.160407 + k / 30, k))
        frames.append((d, c, int(round((1305031102.160407 + k / 30) * 1e6))))
    (tmp_path / "assoc.txt").write_text("\n".join(lines) + "\n")
    r = pkg.FrameReader(tmp_path / "assoc.txt", depth_units_per_metre=5000.0)   # TUM: 5000 units per metre
    assert (r.width, r.height, r.num_frames) == (depth.shape[1], depth.shape[0], 3)
    for d, c, ts in frames:
        got = r.next_host()
        assert got is not None
        assert np.array_equal(got[0], np.rint(d.astype(np.float64) / 5.0).astype(np.uint16))   # -> millimetres
        assert np.array_equal(got[1], c) and got[2] == ts
    assert r.next_host() is None
    r.rewind()
    assert r.next_host()[2] == frames[0][2]
    r.close()


def test_focal_from_fov():
    pkg = load_pkg()
    hf, vf = math.radians(58.5), math.radians(45.6)    # PrimeSense depth FOV
    fx, fy = pkg.focal_from_fov(640, 480, hf, vf)
    assert fx == np.float32(640.0) / (np.float32(2.0) * np.float32(math.tan(np.float32(0.5) * np.float32(hf)))) or abs(fx - 571.4) < 0.5
    assert abs(fx - 640 / (2 * math.tan(hf / 2))) < 1e-3 and abs(fy - 480 / (2 * math.tan(vf / 2))) < 1e-3

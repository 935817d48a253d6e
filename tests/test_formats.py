"""CPU tests of the data formats either side of the hot path: the bincode `rcn.bin` codec, the PNG
front end and the directory-per-class sampling of load_data."""
import os
import struct
import zlib

import numpy as np
import pytest

from mercer_research_amd import checkpoint as ck
from mercer_research_amd import data, png
from mercer_research_amd._lib import RcnPanic


# ------------------------------------------------------------------ rcn.bin (bincode 1.x, rcn.rs:13-25 + serialization.rs)

def _hand_built():
    """Bytes assembled field by field from the serde data model, independently of checkpoint.dumps."""
    b = struct.pack("<Q", 10)                                   # classes
    b += struct.pack("<Q", 2) + struct.pack("<II", 0, 1) + struct.pack("<II", 1, 1)   # [Convolve2D(Same), Pool2D(Max)]
    b += struct.pack("<Q", 1) + struct.pack("<Q", 3)            # feedforward_cfg = [3]
    w0 = np.arange(6, dtype=np.float64).reshape(3, 2) + 0.5      # 3x2
    w1 = -np.arange(30, dtype=np.float64).reshape(10, 3)
    b += struct.pack("<Q", 2)
    for w in (w0, w1):
        b += struct.pack("<QQ", *w.shape) + struct.pack("<Q", w.size) + b"".join(struct.pack("<d", v) for v in w.T.ravel())  # column-major
    b += struct.pack("<Q", 2)
    b0, b1 = np.array([1.0, 2.0, 3.0]), np.linspace(0, 1, 10)
    for v in (b0, b1):
        b += struct.pack("<Q", v.size) + b"".join(struct.pack("<d", x) for x in v)
    b += struct.pack("<dd", 435.4, 547.59)
    for s in ("images/mnist_png/training", "images/mnist_png/testing"):
        b += struct.pack("<Q", len(s)) + s.encode()
    return b, (w0, w1), (b0, b1)


def test_checkpoint_matches_hand_built_bytes():
    raw, ws, bs = _hand_built()
    m = ck.loads(raw)
    assert m.classes == 10 and m.convpool_cfg == [(0, 1), (1, 1)] and m.feedforward_cfg == [3]
    assert all(np.array_equal(a, b) for a, b in zip(m.layer_weights, ws)) and all(np.array_equal(a, b) for a, b in zip(m.layer_bias, bs))
    assert m.scale_set == (435.4, 547.59) and m.training_path.endswith("training") and m.testing_path.endswith("testing")
    assert ck.dumps(m) == raw                                   # byte-exact re-serialisation


def test_checkpoint_empty_model_and_errors():
    fresh = ck.RCNCheckpoint(10, [(0, 1), (1, 1), (0, 1), (1, 1)], [30], training_path="a", testing_path="b")   # RCN::new state
    raw = ck.dumps(fresh)
    assert len(raw) == 8 + 8 + 4 * 8 + 8 + 8 + 8 + 8 + 16 + 8 + 1 + 8 + 1
    back = ck.loads(raw)
    assert back.layer_weights == [] and back.scale_set == (1.0, 1.0)
    with pytest.raises(ck.CheckpointError):
        ck.loads(raw[:-3])                                      # UnexpectedEof
    bad = bytearray(raw); bad[16] = 7                           # variant index 7 of RCNLayer
    with pytest.raises(ck.CheckpointError):
        ck.loads(bytes(bad))
    hb, _, _ = _hand_built()
    broken = bytearray(hb)
    off = 8 + 8 + 16 + 8 + 8 + 8                                # first Weights.dims.0
    broken[off:off + 8] = struct.pack("<Q", 4)                  # dims 4x2 but 6 values
    with pytest.raises(ck.CheckpointError):
        ck.loads(bytes(broken))


# ------------------------------------------------------------------ PNG front end

def _png(width, height, depth, ctype, scanlines, plte=None):
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    if plte is not None:
        out += chunk(b"PLTE", bytes(plte))
    raw = b"".join(scanlines)
    return out + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def _filter_rows(img_bytes, bpp, ftypes):
    """PNG filtering done the slow, obvious way (the inverse of what the decoder must undo)."""
    H, stride = img_bytes.shape
    lines, prev = [], np.zeros(stride, dtype=np.int32)
    for y in range(H):
        cur = img_bytes[y].astype(np.int32)
        ft = ftypes[y % len(ftypes)]
        out = np.zeros(stride, dtype=np.int32)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (cur[i] - pred) & 255
        lines.append(bytes([ft]) + out.astype(np.uint8).tobytes())
        prev = cur
    return lines


def test_png_gray_roundtrip_and_all_filters():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (28, 28)).astype(np.uint8)
    assert np.array_equal(png.to_pixel_matrix_u8(png.encode_gray(img)), img)
    for ftypes in ([1], [2], [3], [4], [0, 1, 2, 3, 4]):
        data_ = _png(28, 28, 8, 0, _filter_rows(img, 1, ftypes))
        assert np.array_equal(png.to_pixel_matrix_u8(data_), img), ftypes


def test_png_colour_types_follow_the_image_crate():
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (5, 7, 3)).astype(np.uint8)
    got = png.to_pixel_matrix_u8(_png(7, 5, 8, 2, _filter_rows(rgb.reshape(5, 21), 3, [4, 1])))
    r32 = rgb.astype(np.uint32)
    want = ((2126 * r32[..., 0] + 7152 * r32[..., 1] + 722 * r32[..., 2]) // 10000).astype(np.uint8)
    assert np.array_equal(got, want)                                           # .grayscale() of Rgb8
    ga = rng.integers(0, 256, (4, 6, 2)).astype(np.uint8)
    assert np.array_equal(png.to_pixel_matrix_u8(_png(6, 4, 8, 4, _filter_rows(ga.reshape(4, 12), 2, [0]))), ga[..., 0])   # LumaA8: alpha ignored
    rgba = rng.integers(0, 256, (3, 3, 4)).astype(np.uint8)
    got = png.to_pixel_matrix_u8(_png(3, 3, 8, 6, _filter_rows(rgba.reshape(3, 12), 4, [3])))
    a32 = rgba.astype(np.uint32)
    assert np.array_equal(got, ((2126 * a32[..., 0] + 7152 * a32[..., 1] + 722 * a32[..., 2]) // 10000).astype(np.uint8))
    bits = np.array([[0b10110000]], dtype=np.uint8)                            # 1-bit gray, width 4: 1 0 1 1 -> 255 0 255 255
    assert np.array_equal(png.to_pixel_matrix_u8(_png(4, 1, 1, 0, _filter_rows(bits, 1, [0]))), np.array([[255, 0, 255, 255]], dtype=np.uint8))
    pal = np.array([[0x01]], dtype=np.uint8)                                   # 4-bit palette indices 0, 1
    got = png.to_pixel_matrix_u8(_png(2, 1, 4, 3, _filter_rows(pal, 1, [0]), plte=[10, 20, 30, 200, 100, 50]))
    assert got.tolist() == [[(2126 * 10 + 7152 * 20 + 722 * 30) // 10000, (2126 * 200 + 7152 * 100 + 722 * 50) // 10000]]
    g16 = np.zeros((2, 4), dtype=np.uint8)
    with pytest.raises(png.InvalidGrayscaleImageError):                        # Luma16 is rejected by get_pixel_matrix
        png.to_pixel_matrix_u8(_png(2, 2, 16, 0, _filter_rows(g16, 2, [0])))
    with pytest.raises(png.PngError):
        png.to_pixel_matrix_u8(b"not a png")


# ------------------------------------------------------------------ directory-per-class sampling (rcn.rs:367-404)

def _mk_set(root, per_class):
    for name, n in per_class.items():
        os.makedirs(os.path.join(root, name))
        for i in range(n):
            img = np.full((6, 6), (hash(name) + i) % 251, dtype=np.uint8)
            with open(os.path.join(root, name, f"{i}.png"), "wb") as f:
                f.write(png.encode_gray(img))


def test_scan_and_sample_semantics(tmp_path):
    root = str(tmp_path / "set")
    _mk_set(root, {"0": 5, "1": 4, "10": 6, "2": 3})
    picked, n_classes = data.scan_and_sample(root, 3, np.random.default_rng(0))
    assert n_classes == 4 and len(picked) == 12
    order = [os.path.basename(os.path.dirname(p)) for p, _ in picked]
    assert order == ["0"] * 3 + ["1"] * 3 + ["10"] * 3 + ["2"] * 3            # string-sorted: "10" before "2" (rcn.rs:374)
    assert [c for _, c in picked] == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3     # class index = position in that order
    assert len({p for p, _ in picked}) == 12                                   # without replacement (paths.remove, rcn.rs:394)
    all3 = {p for p, c in data.scan_and_sample(root, 3, np.random.default_rng(1))[0] if c == 3}
    assert len(all3) == 3                                                      # limit == class size: every file exactly once
    with pytest.raises(RcnPanic) as e:                                         # rcn.rs:383-390
        data.scan_and_sample(root, 4, np.random.default_rng(0))
    assert "too large! expected 4 <= 3" in str(e.value)
    imgs, lab, n = data.load_image_set(root, 2, np.random.default_rng(2))
    assert imgs.shape == (8, 6, 6) and imgs.dtype == np.uint8 and lab.tolist() == [0, 0, 1, 1, 2, 2, 3, 3] and n == 4


# ------------------------------------------------------------------ the C++ host's codecs agree with the Python ones

@pytest.fixture(scope="module")
def format_check(tmp_path_factory):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path_factory.mktemp("fc") / "format_check")
    # AddressSanitizer + UBSan on the CPU build (the GPU pool has no sanitizer runs): every codec test below -- including the
    # truncated / malformed inputs -- runs the C++ decoder under them; any report makes the helper exit non-zero
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    os.path.join(root, "tests", "cpp", "format_check.cpp"), "-lz", "-o", exe], check=True)
    return exe


def test_cpp_bincode_codec_matches_python(format_check, tmp_path):
    import subprocess
    raw, _, _ = _hand_built()
    src, dst = tmp_path / "a.bin", tmp_path / "b.bin"
    src.write_bytes(raw)
    assert subprocess.run([format_check, "bincode", str(src), str(dst)]).returncode == 0
    assert dst.read_bytes() == raw
    src.write_bytes(raw[:-5])
    assert subprocess.run([format_check, "bincode", str(src), str(dst)], capture_output=True).returncode == 4


def test_cpp_png_decoder_matches_python(format_check, tmp_path):
    import subprocess
    rng = np.random.default_rng(5)
    cases = []
    img = rng.integers(0, 256, (28, 28)).astype(np.uint8)
    cases.append(_png(28, 28, 8, 0, _filter_rows(img, 1, [0, 1, 2, 3, 4])))
    rgb = rng.integers(0, 256, (5, 7, 3)).astype(np.uint8)
    cases.append(_png(7, 5, 8, 2, _filter_rows(rgb.reshape(5, 21), 3, [4, 3])))
    ga = rng.integers(0, 256, (4, 6, 2)).astype(np.uint8)
    cases.append(_png(6, 4, 8, 4, _filter_rows(ga.reshape(4, 12), 2, [2])))
    cases.append(_png(4, 1, 1, 0, _filter_rows(np.array([[0b10110000]], dtype=np.uint8), 1, [0])))
    cases.append(_png(2, 1, 4, 3, _filter_rows(np.array([[0x01]], dtype=np.uint8), 1, [0]), plte=[10, 20, 30, 200, 100, 50]))
    cases.append(png.encode_gray(img))
    for i, data_ in enumerate(cases):
        f = tmp_path / f"c{i}.png"
        f.write_bytes(data_)
        out = subprocess.run([format_check, "png", str(f)], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout
        hw, hexpx = out.stdout.strip().split("\n")
        want = png.to_pixel_matrix_u8(data_)
        assert hw == f"{want.shape[0]} {want.shape[1]}" and bytes.fromhex(hexpx) == want.tobytes(), i
    f16 = tmp_path / "g16.png"
    f16.write_bytes(_png(2, 2, 16, 0, _filter_rows(np.zeros((2, 4), dtype=np.uint8), 2, [0])))
    assert subprocess.run([format_check, "png", str(f16)], capture_output=True).returncode == 3


def test_product_and_oracle_synthetic_generators_agree():
    """bench.py draws its workload from mercer_research_amd.synth (the product never imports oracle/); the tests draw from the
    oracle's copy.  Same seeds must give the same arrays."""
    from mercer_research_amd import synth
    from oracle import rcn_oracle as ro
    a, la = synth.synthetic_images(50, seed=7)
    b, lb = ro.synthetic_images(50, seed=7)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    for (wa, ba), (wb, bb) in [(synth.synthetic_params([784, 30, 10], seed=3), ro.synthetic_params([784, 30, 10], seed=3))]:
        assert all(np.array_equal(x, y) for x, y in zip(wa + ba, wb + bb))

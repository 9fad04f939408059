"""CPU tests of the drop-in's I/O side: the cv::imread (PNG) / LoadEXR / cv::remap stand-ins and the
CoFusionReader with the reference's surface, and the drop-in check itself -- the reference's OWN src/main.cpp, unchanged, compiled
against nice-slam-cpp_amd/host/include and linked against libnsk_host.so.

Fixtures are written here from the formats' specifications (PNG: zlib-deflated filtered scanlines; OpenEXR: single-part scanline files,
NONE / RLE / ZIPS / ZIP); nothing is read from the reference's data (it ships none)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nice-slam-cpp_amd", "host")
EXE = os.path.join(HOST, "io_test")


def _png(path, arr, filters=None):
    """arr: uint8 [H,W] / [H,W,3] / [H,W,4] or uint16 [H,W]; every scanline filter type is exercised in turn unless `filters` says otherwise"""
    a = np.asarray(arr)
    H, W = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    bits = 16 if a.dtype == np.uint16 else 8
    ctype = {1: 0, 3: 2, 4: 6}[cn]
    raw = a.astype(">u2").tobytes() if bits == 16 else a.astype(np.uint8).tobytes()
    bpp = cn * bits // 8
    stride = W * bpp
    rows = [bytearray(raw[y * stride:(y + 1) * stride]) for y in range(H)]
    out = bytearray()
    prev = bytearray(stride)
    for y, row in enumerate(rows):
        ft = (filters[y % len(filters)] if filters else y % 5)
        enc = bytearray(stride)
        for x in range(stride):
            a_ = row[x - bpp] if x >= bpp else 0
            b_ = prev[x]
            c_ = prev[x - bpp] if x >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a_
            elif ft == 2:
                p = b_
            elif ft == 3:
                p = (a_ + b_) >> 1
            else:
                pq = a_ + b_ - c_
                pa, pb, pc = abs(pq - a_), abs(pq - b_), abs(pq - c_)
                p = a_ if (pa <= pb and pa <= pc) else (b_ if pb <= pc else c_)
            enc[x] = (row[x] - p) & 255
        out.append(ft)
        out += enc
        prev = row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    comp = zlib.compress(bytes(out))
    idat = chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])      # two IDAT chunks: they must be concatenated
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, bits, ctype, 0, 0, 0)) + chunk(b"tEXt", b"k\x00v") + idat + chunk(b"IEND", b""))


def _exr(path, channels, compression, xmin=0, ymin=0):
    """channels: list of (name, 'half'|'float'|'uint', array [H,W]) -- written in alphabetical order as the format requires"""
    channels = sorted(channels, key=lambda c: c[0])
    H, W = channels[0][2].shape
    ptype = {"uint": 0, "half": 1, "float": 2}

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype[t], 0, 1, 1) for n, t, _ in channels) + b"\0"
    box = struct.pack("<iiii", xmin, ymin, xmin + W - 1, ymin + H - 1)
    hdr = (b"\x76\x2f\x31\x01" + struct.pack("<I", 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression])) +
           attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") +
           attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) +
           attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lpb = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, H, lpb):
        raw = b""
        for y in range(y0, min(H, y0 + lpb)):
            for n, t, a in channels:
                raw += a[y].astype({"half": "<f2", "float": "<f4", "uint": "<u4"}[t]).tobytes()
        if compression == 0:
            data = raw
        else:
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)             # interleave into two halves
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256                          # predictor
            pre = d.astype(np.uint8).tobytes()
            if compression == 1:                                                # RLE
                o, i = bytearray(), 0
                while i < len(pre):
                    j = i
                    while j + 1 < len(pre) and pre[j + 1] == pre[i] and j - i < 126:
                        j += 1
                    if j - i >= 2:
                        o += struct.pack("b", j - i) + pre[i:i + 1]
                        i = j + 1
                    else:
                        k = i
                        while k < len(pre) and k - i < 127 and not (k + 2 < len(pre) and pre[k] == pre[k + 1] == pre[k + 2]):
                            k += 1
                        o += struct.pack("b", -(k - i)) + pre[i:k]
                        i = k
                data = bytes(o)
            else:
                data = zlib.compress(pre)
            if len(data) >= len(raw):
                data = raw                                                      # the format stores a block raw when compression does not pay
        blocks.append((y0 + ymin, data))
    table_pos = len(hdr)
    pos = table_pos + 8 * len(blocks)
    offs, body = [], b""
    for y, data in blocks:
        offs.append(pos)
        blk = struct.pack("<ii", y, len(data)) + data
        body += blk
        pos += len(blk)
    with open(path, "wb") as f:
        f.write(hdr + b"".join(struct.pack("<Q", o) for o in offs) + body)


POSED = None


def _rand_pose(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4); m[:3, :3] = R; m[:3, 3] = rng.uniform(-2, 2, 3)
    return m, q


def _write_posed_sequences(d, rng):
    """tiny Replica / ScanNet / TUM-RGBD sequences in the upstream NICE-SLAM layouts (PNG colour); returns what the readers must hand out"""
    want = {}
    H, W = 6, 8
    # Replica: results/frame%06d.png, results/depth%06d.png (/6553.5), traj.txt
    rd = os.path.join(d, "replica"); os.makedirs(os.path.join(rd, "results"))
    fr = []
    with open(os.path.join(rd, "traj.txt"), "w") as f:
        for i in range(3):
            col = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); dep = rng.integers(0, 65536, (H, W)).astype(np.uint16); m, _ = _rand_pose(rng)
            _png(os.path.join(rd, "results", "frame%06d.png" % i), col); _png(os.path.join(rd, "results", "depth%06d.png" % i), dep)
            f.write(" ".join("%.9g" % v for v in m.reshape(-1)) + "\n")
            fr.append((col, dep.astype(np.float32) / np.float32(6553.5), m))
    want["replica"] = fr
    # a Replica frame whose colour image is a JPEG: the stand-in must refuse it with a clear message
    rj = os.path.join(d, "replica_jpg"); os.makedirs(os.path.join(rj, "results"))
    _png(os.path.join(rj, "results", "depth000000.png"), np.zeros((H, W), np.uint16))
    open(os.path.join(rj, "results", "frame000000.jpg"), "wb").write(b"\xff\xd8\xff\xe0 not decoded here")
    open(os.path.join(rj, "traj.txt"), "w").write("1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1\n")
    # ScanNet: color/%d.png, depth/%d.png (/1000), pose/%d.txt
    sd = os.path.join(d, "scannet")
    for sub in ("color", "depth", "pose"):
        os.makedirs(os.path.join(sd, sub))
    fr = []
    for i in range(2):
        col = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); dep = rng.integers(0, 8000, (H, W)).astype(np.uint16); m, _ = _rand_pose(rng)
        _png(os.path.join(sd, "color", "%d.png" % i), col); _png(os.path.join(sd, "depth", "%d.png" % i), dep)
        open(os.path.join(sd, "pose", "%d.txt" % i), "w").write("\n".join(" ".join("%.9g" % v for v in row) for row in m) + "\n")
        fr.append((col, dep.astype(np.float32) / np.float32(1000.0), m))
    want["scannet"] = fr
    # TUM: stamped lists; rgb frames at 0.00, 0.03 (too close at 10 Hz: dropped), 0.15, 0.30 (no depth within 0.08 s: dropped), 0.45
    td = os.path.join(d, "tum"); os.makedirs(os.path.join(td, "rgb")); os.makedirs(os.path.join(td, "depth"))
    stamps = [0.00, 0.03, 0.15, 0.30, 0.45]
    dstamps = [0.01, 0.04, 0.16, 0.40, 0.46]
    cols, deps, gts = [], [], []
    with open(os.path.join(td, "rgb.txt"), "w") as f:
        f.write("# color images\n# timestamp filename\n")
        for t in stamps:
            col = rng.integers(0, 256, (H, W, 3), dtype=np.uint8); cols.append(col)
            _png(os.path.join(td, "rgb", "%.6f.png" % t), col); f.write("%.6f rgb/%.6f.png\n" % (t, t))
    with open(os.path.join(td, "depth.txt"), "w") as f:
        f.write("# depth maps\n")
        for t in dstamps:
            dep = rng.integers(0, 30000, (H, W)).astype(np.uint16); deps.append(dep)
            _png(os.path.join(td, "depth", "%.6f.png" % t), dep); f.write("%.6f depth/%.6f.png\n" % (t, t))
    with open(os.path.join(td, "groundtruth.txt"), "w") as f:
        f.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
        for t in [0.005, 0.035, 0.155, 0.305, 0.455]:
            m, q = _rand_pose(rng); gts.append(m)
            f.write("%.6f %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n" % ((t,) + tuple(m[:3, 3]) + tuple(q)))
    keep = [0, 2, 4]                                             # rgb indices that survive; their depth / pose partners have the same index
    inv0 = np.linalg.inv(gts[0])
    want["tum"] = [(cols[i], deps[i].astype(np.float32) / np.float32(5000.0), inv0 @ gts[i]) for i in keep]
    return want


@pytest.fixture(scope="module")
def io_dump(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", HOST, "all"])
    d = str(tmp_path_factory.mktemp("io"))
    rng = np.random.default_rng(0)
    fx = {"rgb8": rng.integers(0, 256, (7, 5, 3), dtype=np.uint8), "rgba8": rng.integers(0, 256, (6, 9, 4), dtype=np.uint8),
          "gray8": rng.integers(0, 256, (11, 4), dtype=np.uint8), "gray16": rng.integers(0, 65536, (5, 6)).astype(np.uint16)}
    for k, a in fx.items():
        _png(os.path.join(d, k + ".png"), a)
    open(os.path.join(d, "not_a.png"), "wb").write(b"hello")
    depth = rng.uniform(0.3, 5.0, (37, 23)).astype(np.float32)                  # 37 rows: ZIP blocks of 16, 16 and 5 lines
    smooth = (np.add.outer(np.arange(37), np.arange(23)) * 0.125).astype(np.float32)   # compressible: run-length and zlib paths really compress
    ex = {"f_none": ([("Z", "float", depth)], 0), "f_zip": ([("Y", "float", smooth)], 3), "f_zips": ([("R", "float", depth)], 2),
          "h_zip_rgba": ([("R", "half", smooth), ("G", "half", smooth * 2), ("B", "half", depth), ("A", "half", np.ones_like(depth))], 3),
          "f_rle": ([("Z", "float", np.floor(smooth))], 1), "u_none": ([("Z", "uint", np.floor(depth * 10))], 0)}
    for k, (ch, comp) in ex.items():
        _exr(os.path.join(d, k + ".exr"), ch, comp, xmin=2, ymin=-3)
    seq = os.path.join(d, "seq")
    os.makedirs(os.path.join(seq, "colour")); os.makedirs(os.path.join(seq, "depth_noise"))
    seq_fx = {}
    for idx in (3, 4):                                                          # src/inputs/CoFusionReader.cpp:40-41 file names
        col = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
        dep = rng.uniform(0.5, 4.0, (6, 8)).astype(np.float32)
        _png(os.path.join(seq, "colour", "Color0%03d.png" % idx), col)
        _exr(os.path.join(seq, "depth_noise", "Depth0%03d.exr" % idx), [("R", "float", dep), ("G", "float", dep * 2), ("B", "float", dep * 3)], 3)
        seq_fx[idx] = (col, dep)
    global POSED
    POSED = _write_posed_sequences(d, rng)
    r = subprocess.run([EXE, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {f[4:-4]: np.load(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("out_")}
    return out, fx, ex, seq_fx, d


def test_readers_survive_truncated_and_corrupted_files(io_dump, tmp_path):
    """the PNG / EXR readers under AddressSanitizer + UBSan on every truncation and thousands of byte corruptions of the valid fixtures: they must
    fail cleanly (empty Mat / error code) or decode, never read out of bounds (make io_fuzz_asan; the run aborts on the first sanitizer report)"""
    subprocess.check_call(["make", "-s", "-C", HOST, "io_fuzz_asan"])
    r = subprocess.run([os.path.join(HOST, "io_fuzz_asan"), io_dump[4]], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1"))
    assert r.returncode == 0 and "io_fuzz ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_png_reader(io_dump):
    out, fx, _, _ = io_dump[:4]
    assert np.array_equal(out["rgb8"], fx["rgb8"][:, :, ::-1].astype(np.float32))            # OpenCV order: B,G,R
    assert np.array_equal(out["rgba8"], fx["rgba8"][:, :, [2, 1, 0, 3]].astype(np.float32))
    assert np.array_equal(out["gray8"], fx["gray8"].astype(np.float32))
    assert np.array_equal(out["gray16"], fx["gray16"].astype(np.float32))
    assert out["gray8_color"].shape == (11, 4, 3) and np.array_equal(out["gray8_color"][:, :, 1], fx["gray8"].astype(np.float32))


def test_exr_reader(io_dump):
    out, _, ex, _ = io_dump[:4]
    for k, (chans, comp) in ex.items():
        got = out[k]
        H, W = chans[0][2].shape
        assert got.shape == (H, W, 4)
        names = {n: (t, a) for n, t, a in chans}
        if len(chans) == 1:                                                     # a single channel lands in R, G and B
            t, a = chans[0][1], chans[0][2]
            want = a.astype(np.float16).astype(np.float32) if t == "half" else a.astype(np.float32)
            for c in range(3):
                assert np.array_equal(got[:, :, c], want), k
            assert (got[:, :, 3] == 1).all()
        else:
            for c, n in enumerate("RGBA"):
                t, a = names[n]
                assert np.array_equal(got[:, :, c], a.astype(np.float16).astype(np.float32)), (k, n)


def test_remap_bilinear_zero_border(io_dump):
    out, _, _, _ = io_dump[:4]
    src = (np.arange(12, dtype=np.float32) ** 2).reshape(3, 4)
    xs = [0.0, 1.5, 2.25, 3.0, -0.5, 3.5]
    ys = [0.0, 0.5, 1.75, 2.0, 1.0, 2.5]

    def px(y, x):
        return 0.0 if (x < 0 or x >= 4 or y < 0 or y >= 3) else float(src[y, x])
    want = []
    for u, v in zip(xs, ys):
        x0, y0 = int(np.floor(u)), int(np.floor(v))
        ax, ay = u - x0, v - y0
        want.append((1 - ax) * (1 - ay) * px(y0, x0) + ax * (1 - ay) * px(y0, x0 + 1) + (1 - ax) * ay * px(y0 + 1, x0) + ax * ay * px(y0 + 1, x0 + 1))
    assert np.allclose(out["remap"].ravel(), np.array(want, np.float32), rtol=1e-6, atol=1e-6)


def test_cofusion_reader(io_dump):
    """include/inputs/CoFusionReader.h surface: frames 3 and 4 of a tiny sequence; depth = channel 0 of the EXR (D27), colour / 255 in
    OpenCV's B,G,R order (src/inputs/CoFusionReader.cpp:44-51 as written)"""
    out, _, _, seq = io_dump[:4]
    for idx, (col, dep) in seq.items():
        assert np.array_equal(out["seq_depth%d" % idx], dep)
        assert np.allclose(out["seq_rgb%d" % idx], col[:, :, ::-1].astype(np.float32) / 255.0, atol=1e-7)


def test_reference_main_compiles_unchanged_against_the_drop_in_headers():
    """north_star: "drops into src/main.cpp unchanged".  The reference's own src/main.cpp (it includes only "Tracker.h") is compiled
    against host/include -- Tracker / NICE / CoFusionReader class surface, the yaml-cpp / OpenCV / Eigen / tinyexr stand-ins, libtorch as
    the host tensor container -- and linked against libnsk_host.so.  It is not run here (it wants a GPU and absolute dataset paths)."""
    ref = "/root/reference/src/main.cpp"
    if not os.path.exists(ref):
        pytest.skip("the reference is not mounted on this box")
    main_ref = os.path.join(HOST, "main_ref")
    if os.path.exists(main_ref):
        os.remove(main_ref)
    r = subprocess.run(["make", "-C", HOST, "main_ref"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(main_ref), r.stdout[-3000:] + r.stderr[-3000:]
    # the binary really references the drop-in's symbols
    nm = subprocess.run(["nm", "-D", "--undefined-only", "-C", main_ref], capture_output=True, text=True).stdout
    for sym in ("Tracker::run(CoFusionReader&, NICE)", "Tracker::Tracker(", "NICE::NICE(", "CoFusionReader::CoFusionReader("):
        assert sym in nm, sym


@pytest.mark.parametrize("name", ["replica", "scannet", "tum"])
def test_posed_sequence_readers(io_dump, name):
    """inputs/SequenceReader.h (the readers BASELINE.json's K2-K5 presuppose and the reference lacks): colour / 255 in B,G,R order, depth in
    metres (png / 6553.5, 1000, 5000), poses as OpenGL cameras (columns 1 and 2 negated); TUM: rgb stamps associated with depth and pose
    within 0.08 s, thinned to the frame rate, poses relative to the first frame"""
    out = io_dump[0]
    want = POSED[name]
    poses = out[name + "_poses"]
    assert poses.shape == (len(want), 4, 4)
    for i, (col, dep, m) in enumerate(want):
        assert np.allclose(out["%s_rgb%d" % (name, i)], col[:, :, ::-1].astype(np.float32) / 255.0, atol=1e-7)
        assert np.allclose(out["%s_depth%d" % (name, i)], dep, rtol=1e-6, atol=1e-7)
        gl = m.copy(); gl[:3, 1] *= -1; gl[:3, 2] *= -1
        assert np.allclose(poses[i], gl, atol=2e-6), (name, i)

"""dfd_analyze_jpegs_host on 256 natural-texture 1080p JPEGs (chunks of 64): wall time per call; under rocprofv3 the kernel
trace says where a chunk's time goes (decode kernels vs the analysis)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
if os.environ.get("JP_TORCH"):
    import torch
    if os.environ["JP_TORCH"] == "2":
        torch.manual_seed(0); _ = torch.randn(256, 3, 224, 224) @ torch.randn(224, 224)
from PIL import Image
import rtdfd_amd
import frames as F

W = rtdfd_amd.weights
K = 4
h = rtdfd_amd._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=256)
h.warmup(256, 64)
if os.environ.get("JP_CHUNK"): h.set_option("jpeg_chunk_bytes", int(os.environ["JP_CHUNK"]))
kind = os.environ.get("JP_KIND", "natural")
def bench_natural(seed, H=1080, W=1920):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), np.float32)
    for c in range(3):
        img[..., c] = 120 + 60 * np.sin(xx / (190.0 + 23 * c) + c) * np.cos(yy / (140.0 - 11 * c)) + 25 * np.sin((xx + 2 * yy) / 37.0)
    low = r.normal(0, 1, (H // 8 + 1, W // 8 + 1, 3)).astype(np.float32)
    img += 14 * np.kron(low, np.ones((8, 8, 1), np.float32))[:H, :W]
    img += r.normal(0, 3.0, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)

files = []
for i in range(8):
    fr = (np.random.default_rng(7 + i).integers(50, 200, (1080, 1920, 3), dtype=np.uint8) if kind == "noise" else
          bench_natural(40 + i) if kind == "bench" else F.natural_like(1080, 1920, seed=9 + i))
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(fr[..., ::-1])).save(buf, format="JPEG", quality=85)
    files.append(buf.getvalue())
n = int(os.environ.get("JP_N", "256"))
datas = [files[i % 8] for i in range(n)]
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * n
packed = h.pack_jpegs(datas)
if os.environ.get("JP_BURN"):
    # sustained-clock comparison: the resident analysis of the same decoded frames, timed in this process after a burn-in
    dec = h.decode_jpeg_batch(datas[:64])
    fd = h.alloc(dec.nbytes).upload(dec)
    t_end = time.perf_counter() + float(os.environ["JP_BURN"])
    while time.perf_counter() < t_end:
        h.analyze_batch_device(fd.ptr, 64, 1080, 1920, forced_boxes=boxes[:64], max_faces=K)
    t0 = time.perf_counter()
    for _ in range(8):
        h.analyze_batch_device(fd.ptr, 64, 1080, 1920, forced_boxes=boxes[:64], max_faces=K)
    dt = (time.perf_counter() - t0) / 8
    print("resident analysis of 64 decoded frames: %.2f ms = %.0f frames/s" % (dt * 1e3, 64 / dt), flush=True)
    fd.free()
for _ in range(2):
    h.analyze_jpegs_host(datas, 64, forced_boxes=boxes, max_faces=K, packed=packed)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    h.analyze_jpegs_host(datas, 64, forced_boxes=boxes, max_faces=K, packed=packed)
    ts.append(time.perf_counter() - t0)
print(kind, "bytes/frame", len(files[0]), "ms per call", [round(t * 1e3, 2) for t in ts], "frames/s", round(n / sorted(ts)[1]))

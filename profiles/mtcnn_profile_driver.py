"""The MTCNN-on batch path alone (selective seeded cascade) for rocprofv3 --stats / wall-clock comparison."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd as pkg
W = pkg.weights
sel = os.environ.get("MT_DENSE", "0") != "1"
h = pkg._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0), W.seeded_mtcnn_state_dict(0, W.MTCNN_SELECTIVE if sel else None)), device=0, max_batch=4 * int(os.environ.get('MT_FRAMES', '8')))
NF = int(os.environ.get("MT_FRAMES", "8"))
frames = np.random.default_rng(7).integers(50, 200, (NF, 1080, 1920, 3), dtype=np.uint8)
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * NF
h.warmup(4 * NF, NF)                                        # measured GEMM tiles, as in bench.py
fd = h.alloc(frames.nbytes).upload(frames)
if os.environ.get('MT_OFF') == '1':
    h.set_option('mtcnn', 0)
h.analyze_batch_device(fd.ptr, NF, 1080, 1920, forced_boxes=boxes, max_faces=4)
h.sync()
t0 = time.perf_counter()
its = []
for _ in range(int(os.environ.get("MT_ITERS", "5"))):
    t1 = time.perf_counter()
    res = h.analyze_batch_device(fd.ptr, NF, 1080, 1920, forced_boxes=boxes, max_faces=4)
    its.append(round((time.perf_counter() - t1) * 1e3, 2))
h.sync()
dt = (time.perf_counter() - t0) / len(its)
print("per call ms", its)
flat = np.concatenate([np.asarray(l).reshape(-1) for l in res[1]])
print(f"wall per call {dt*1e3:.2f} ms for {4*NF} crops ({dt/(4*NF)*1e3:.3f} ms/crop); crops with a face {int((~np.isnan(flat)).sum())}/{4*NF}")

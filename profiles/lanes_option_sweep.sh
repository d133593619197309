# Probe (GPU box, repo root): the classifier step of bench.py with two forwards in flight under the library option switches.
run() { python bench.py --steps 40 --warmup 5 --no-e2e --no-streams --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'], j['one_forward_in_flight']['crops_per_s'])"; }
run base
DFD_FUSE_LATE=0 run fuse_late0
DFD_FUSE_LATE_SKIP=0 run skip0
DFD_FUSE_LATE_SKIP=1792 run skip8_9_10
DFD_FUSE_LATE_SKIP=256 run skip8
DFD_SE_IN_PROJ=1 run se_in_proj
DFD_FUSE_SE=1 run fuse_se
DFD_BENCH_LANES=4 run lanes4
run base

import sys, os, json, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import reflectance_filtering_amd as rf
from reflectance_filtering_amd import _ffi
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda", 0)
n, h, w = 64, 1080, 1920
scene, grey = bench.synth_batch(torch, n, h, w, 3234, dev)
dst = torch.empty_like(grey)
g1 = grey[..., :1].contiguous(); d1 = torch.empty_like(g1); g1j = g1.clone()
def t(f, reps=7):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
mp = n*h*w/1e6
a = t(lambda: rf.ops.joint_bilateral_u8(scene, grey, -1, 20.0, 22.0, out=dst))
b = t(lambda: rf.ops.joint_bilateral_u8(g1j, g1, -1, 20.0, 22.0, out=d1, grey_as_bgr=True))
c = t(lambda: rf.ops.joint_bilateral_u8(scene[:8], grey[:8], -1, 20.0, 36.0, out=dst[:8]))
print(json.dumps({"lib": _ffi.LIB_PATH[-12:], "headline": round(mp/(a*1e-3),1), "bf_cnn_cnn": round(mp/(b*1e-3),1), "s36": round(mp/8/(c*1e-3),1)}))

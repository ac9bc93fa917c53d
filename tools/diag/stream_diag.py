import ctypes, os, sys, subprocess
sys.path.insert(0, os.getcwd())
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch
    import opencorr_amd as oc
else:
    import opencorr_amd as oc
    oc.capi.lib()
    import torch
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "librccl" in l or "hsa-runtime" in l})
print(order, maps(), flush=True)
hip = oc.capi.hip_runtime()
print("after ctypes", maps(), flush=True)
from opencorr_amd import synth
ref, tar = synth.speckle_pair_2d(128, 128, seed=1)
f = oc.FFTCC2D(16, 16)
f.set_images(ref, tar)
step = sys.argv[2] if len(sys.argv) > 2 else "torch_stream"
if step == "torch_stream":
    s = torch.cuda.Stream()
    print("torch stream", hex(s.cuda_stream), flush=True)
    f.set_stream(s.cuda_stream)
    print("set_stream(torch stream) ok", flush=True)
else:
    hip.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    h = ctypes.c_void_p()
    print("create", hip.hipStreamCreate(ctypes.byref(h)), hex(h.value), flush=True)
    f.set_stream(h.value)
    print("set_stream(ctypes stream) ok", flush=True)
f.reset_stream()
print("done", flush=True)

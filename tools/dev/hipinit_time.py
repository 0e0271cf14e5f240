import ctypes, os, sys, time
t0 = time.time()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../vgan_amd/lib/libvgan_gpu.so"))
t1 = time.time()
lib.vgan_device_warmup(0)
t2 = time.time()
print("dlopen %.0f ms, vgan_device_warmup (hipGetDeviceCount + hipSetDevice + hipFree(0)) %.0f ms  env=%s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "GPU_", "AMD_", "ROC"))}))

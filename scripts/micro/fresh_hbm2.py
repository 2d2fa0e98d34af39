"""hipMalloc of `count` blocks of `gb` GB in a fresh process, timed in groups of 16 GB: python fresh_hbm2.py gb count"""
import ctypes
import sys
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
gb, count = float(sys.argv[1]), int(sys.argv[2])
size = int(gb * (1 << 30))
assert hip.hipSetDevice(0) == 0
w = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(w), 1 << 20) == 0 and hip.hipMemset(w, 0, 1 << 20) == 0 and hip.hipDeviceSynchronize() == 0
times, t_all = [], time.perf_counter()
for i in range(count):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), size)
    times.append(time.perf_counter() - t0)
    if rc != 0:
        print("hipMalloc failed at block", i, rc)
        break
total = time.perf_counter() - t_all
print(f"{count} x {gb} GB: {total * 1e3:.1f} ms in all; per block ms:", " ".join(f"{t * 1e3:.0f}" for t in times))

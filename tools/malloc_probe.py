import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes=[C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes=[C.c_void_p]
hip.hipSetDevice(0)
p=C.c_void_p(); hip.hipMalloc(C.byref(p), 1<<20); hip.hipFree(p)
for rep in range(2):
    for gb in (0.5, 1, 2, 4, 6):
        t=time.perf_counter(); c0=time.process_time(); rc=hip.hipMalloc(C.byref(p), int(gb*(1<<30))); t1=time.perf_counter(); c1=time.process_time()
        hip.hipFree(p); t2=time.perf_counter()
        print("hipMalloc %.1f GB: %.1f ms (cpu %.1f ms)  hipFree %.1f ms  rc %d" % (gb, (t1-t)*1e3, (c1-c0)*1e3, (t2-t1)*1e3, rc))

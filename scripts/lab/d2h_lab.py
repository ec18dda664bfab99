"""How fast can 512 MB leave the device?  pageable hipMemcpy vs hipHostMalloc'ed target vs hipHostRegister."""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
NB = 512 << 20
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), NB) == 0
hip.hipMemset(d, 1, NB); hip.hipDeviceSynchronize()
for rep in range(2):
    a = np.empty(NB, np.uint8)
    t = time.time(); hip.hipMemcpy(a.ctypes.data, d, NB, 2); t1 = time.time() - t
    t = time.time(); hip.hipMemcpy(a.ctypes.data, d, NB, 2); t2 = time.time() - t
    print(f"pageable fresh numpy: {t1*1e3:.1f} ms, again (pages touched): {t2*1e3:.1f} ms")
    p = C.c_void_p()
    t = time.time(); assert hip.hipHostMalloc(C.byref(p), NB, 0) == 0; ta = time.time() - t
    t = time.time(); hip.hipMemcpy(p, d, NB, 2); tc = time.time() - t
    t = time.time(); hip.hipMemcpy(p, d, NB, 2); tc2 = time.time() - t
    v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(NB,))
    t = time.time(); s = int(v[::4096].sum()); tr = time.time() - t
    t = time.time(); hip.hipHostFree(p); tf = time.time() - t
    print(f"hipHostMalloc {ta*1e3:.1f} ms, copy {tc*1e3:.1f} / {tc2*1e3:.1f} ms, touch {tr*1e3:.1f} ms, free {tf*1e3:.1f} ms")
    b = np.empty(NB, np.uint8)
    t = time.time(); rc = hip.hipHostRegister(b.ctypes.data, NB, 0); tr = time.time() - t
    t = time.time(); hip.hipMemcpy(b.ctypes.data, d, NB, 2); tc = time.time() - t
    t = time.time(); hip.hipHostUnregister(b.ctypes.data); tu = time.time() - t
    print(f"hipHostRegister rc={rc} {tr*1e3:.1f} ms, copy {tc*1e3:.1f} ms, unregister {tu*1e3:.1f} ms")
    b = np.zeros(NB, np.uint8)
    t = time.time(); rc = hip.hipHostRegister(b.ctypes.data, NB, 0); tr = time.time() - t
    t = time.time(); hip.hipMemcpy(b.ctypes.data, d, NB, 2); tc = time.time() - t
    hip.hipHostUnregister(b.ctypes.data)
    print(f"hipHostRegister (touched pages) {tr*1e3:.1f} ms, copy {tc*1e3:.1f} ms")
    # chunked through a small pinned buffer + memcpy out
    CH = 32 << 20
    q = C.c_void_p(); hip.hipHostMalloc(C.byref(q), CH, 0)
    out = np.empty(NB, np.uint8)
    qv = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint8)), shape=(CH,))
    t = time.time()
    for off in range(0, NB, CH):
        hip.hipMemcpy(q, C.c_void_p(d.value + off), CH, 2)
        out[off:off + CH] = qv
    print(f"chunked via 32 MB pinned + numpy copy: {(time.time()-t)*1e3:.1f} ms")
    hip.hipHostFree(q)

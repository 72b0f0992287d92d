"""Host readers over the C ABI (no GPU needed): GAF tokenizer (rcls.rs:119-146) and the species graph
loaders (read_gfa profile.rs:466-545, bincode `.bin` zip.rs:236-247)."""
import ctypes as C

import numpy as np

from . import _ffi


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()


def load_gaf(path, n_threads=4, engine=None):
    """-> dict(step_off, node_id, pstart, pend, qlen, mapq, flags) of numpy arrays (packed layout).
    engine=None: the host tokenizer; an Engine: the same tokenisation on its GPU (pantax_hip_gaf_load_device)."""
    lib = _ffi.load()
    h = C.c_void_p()
    err = C.c_char_p()
    if engine is not None:
        rc = lib.pantax_hip_gaf_load_device(engine.ctx, str(path).encode(), C.byref(h))
        if rc != 0:
            raise _ffi.PantaxHipError(rc, lib.pantax_hip_last_error(engine.ctx).decode())
    else:
        rc = lib.pantax_hip_gaf_load(str(path).encode(), n_threads, C.byref(h), C.byref(err))
        if rc != 0:
            raise _ffi.PantaxHipError(rc, (err.value or b"").decode())
    try:
        v = _ffi.PackedReads()
        lib.pantax_hip_gaf_view(h, C.byref(v))
        R, T = v.n_reads, v.n_steps
        return dict(step_off=_arr(v.step_off, R + 1, np.uint32), node_id=_arr(v.node_id, T, np.uint32),
                    pstart=_arr(v.pstart, R, np.uint32), pend=_arr(v.pend, R, np.uint32), qlen=_arr(v.qlen, R, np.uint32),
                    mapq=_arr(v.mapq, R, np.uint8), flags=_arr(v.flags, R, np.uint8))
    finally:
        lib.pantax_hip_gaf_free(h)


def load_graph(path, fmt="gfa"):
    """-> (node_len int64 [V], hap_names [H] (byte order), path_off uint64 [H+1], path_nodes uint32 [P])"""
    lib = _ffi.load()
    h = C.c_void_p()
    err = C.c_char_p()
    rc = lib.pantax_hip_graph_load(str(path).encode(), {"gfa": 0, "bin": 1, "lz4": 2, "zst": 3}[fmt], C.byref(h), C.byref(err))
    if rc != 0:
        raise _ffi.PantaxHipError(rc, (err.value or b"").decode())
    try:
        nn, nh = C.c_uint64(), C.c_uint64()
        nl, po, pn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        names = C.POINTER(C.c_char_p)()
        lib.pantax_hip_graph_view(h, C.byref(nn), C.byref(nh), C.byref(nl), C.byref(po), C.byref(pn), C.byref(names))
        H = nh.value
        path_off = _arr(po, H + 1, np.uint64)
        return (_arr(nl, nn.value, np.int64), [names[i].decode() for i in range(H)], path_off,
                _arr(pn, int(path_off[-1]) if H else 0, np.uint32))
    finally:
        lib.pantax_hip_graph_free(h)

"""ctypes binding of libtnml_host.so: the host-side pieces of the fixedL driver (input-file parser,
idx-ubyte reader, TNMLW1 weight files, initial-W builder) behind a small C API, no GPU involved."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NL = 10


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libtnml_host.so")
        if not os.path.exists(path):
            raise ImportError(path + " is missing: run `make -C tnml_amd/host`")
        L = C.CDLL(path)
        L.tnmlh_last_error.restype = C.c_char_p
        L.tnmlh_input_get.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.tnmlh_input_yesno.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.tnmlh_read_mnist.argtypes = [C.c_char_p, C.c_int, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_ubyte), C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.tnmlh_build_initial_w.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_ulonglong, C.c_char_p,
                                            C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_double]
        L.tnmlh_reduce.argtypes = [C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.tnmlh_build_initial_single.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_char_p, C.c_int, C.c_double]
        L.tnmlh_sites_write.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.tnmlh_sites_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tnmlh_mps_info.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.tnmlh_mps_site.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_double)]
        L.tnmlh_mps_write.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def _err():
    return RuntimeError(load().tnmlh_last_error().decode())


def input_get(path, key):
    buf = C.create_string_buffer(1024)
    rc = load().tnmlh_input_get(path.encode(), key.encode(), buf, 1024)
    if rc < 0:
        raise _err()
    return None if rc == 1 else buf.value.decode()


def input_yesno(path, key, default=False):
    rc = load().tnmlh_input_yesno(path.encode(), key.encode(), int(default))
    if rc < 0:
        raise _err()
    return bool(rc)


def read_mnist(datadir, train=True, nt_per_label=60000):
    L = load()
    n, npix = C.c_int(), C.c_int()
    if L.tnmlh_read_mnist(datadir.encode(), int(train), nt_per_label, n, npix, None, None, None) != 0:
        raise _err()
    px = np.empty((n.value, npix.value), dtype=np.uint8)
    lab = np.empty(n.value, dtype=np.int32)
    idx = np.empty(n.value, dtype=np.int64)
    if L.tnmlh_read_mnist(datadir.encode(), int(train), nt_per_label, n, npix, px.ctypes.data_as(C.POINTER(C.c_ubyte)),
                          lab.ctypes.data_as(C.POINTER(C.c_int)), idx.ctypes.data_as(C.POINTER(C.c_long))) != 0:
        raise _err()
    return px, lab, idx


def read_mps(path):
    """TNMLW1 file -> list of numpy arrays A_j[l,s,r(,L)]"""
    L = load()
    N, c0 = C.c_int(), C.c_int()
    if L.tnmlh_mps_info(path.encode(), N, c0) != 0:
        raise _err()
    W = []
    for j in range(1, N.value + 1):
        ml, mr, Ld = C.c_int(), C.c_int(), C.c_int()
        if L.tnmlh_mps_site(path.encode(), j, ml, mr, Ld, None) != 0:
            raise _err()
        shape = (ml.value, 2, mr.value) + ((NL,) if Ld.value == NL else ())
        buf = np.empty(int(np.prod(shape)))
        L.tnmlh_mps_site(path.encode(), j, ml, mr, Ld, buf.ctypes.data_as(C.POINTER(C.c_double)))
        W.append(buf.reshape(shape, order="F"))
    return W


def write_mps(path, W):
    dims = np.array([[A.shape[0], A.shape[2], NL if A.ndim == 4 else 1] for A in W], dtype=np.int32).ravel()
    data = np.concatenate([np.asarray(A, dtype=np.float64).ravel(order="F") for A in W])
    if load().tnmlh_mps_write(path.encode(), len(W), dims.ctypes.data_as(C.POINTER(C.c_int)),
                              data.ctypes.data_as(C.POINTER(C.c_double))) != 0:
        raise _err()


def build_initial_w(datadir, nt_per_label, ninitial, seed, out, imglen=0, feature_scale=1.0):
    ovl, md = C.c_double(), C.c_int()
    if load().tnmlh_build_initial_w(datadir.encode(), nt_per_label, ninitial, seed, out.encode(), ovl, md, imglen, feature_scale) != 0:
        raise _err()
    return ovl.value, md.value


def reduce(pixels, side, newlen):
    """block-mean down-sampling of [n, side*side] uint8 images to [n, newlen*newlen] real values (byte units)"""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    n = pixels.shape[0]
    out = np.zeros((n, newlen * newlen))
    if load().tnmlh_reduce(pixels.ctypes.data_as(C.POINTER(C.c_ubyte)), n, side, newlen, out.ctypes.data_as(C.POINTER(C.c_double))) != 0:
        raise _err()
    return out


def build_initial_single(datadir, nt_per_label, label, ninitial, seed, normal, out, imglen=0, feature_scale=1.0):
    """initial W of the per-label variant (single.cc:112-128) written to `out`"""
    if load().tnmlh_build_initial_single(datadir.encode(), nt_per_label, label, ninitial, seed, int(normal), out.encode(), imglen, feature_scale) != 0:
        raise _err()


def write_sites(path, N, d=2):
    if load().tnmlh_sites_write(path.encode(), N, d) != 0:
        raise _err()


def read_sites(path):
    N, d = C.c_int(), C.c_int()
    if load().tnmlh_sites_read(path.encode(), N, d) != 0:
        raise _err()
    return N.value, d.value

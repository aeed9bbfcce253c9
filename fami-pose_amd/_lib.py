"""ctypes binding of libfami_hip.so (C ABI declared in include/fami.h).

The signatures are read from the header itself so the binding cannot drift from
the declaration.  There is NO fallback: if the shared library is missing the
import of any compute entry point raises, and every call checks the return code.
"""
import ctypes
import os
import re

from . import options
import torch  # loads torch's HIP runtime first; libfami_hip.so binds to the same libamdhip64 (soname match)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libfami_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'fami.h')

_SCALAR = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
           'fami_stream_t': ctypes.c_void_p}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every fami_* prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int|long)\s+(fami_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if '*' in ret else _SCALAR[ret.strip()]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    toks = a.replace('const', ' ').split()
                    argtypes.append(_SCALAR[toks[0]])
        protos[name] = (restype, argtypes)
    return protos


class FamiError(RuntimeError):
    pass


ROUTE_HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'fami_route.h')


def _route_struct(path=ROUTE_HEADER_PATH):
    """ctypes mirror of fami_route_t, built from the header's own field list (ints and longs only)."""
    src = open(path).read()
    body = re.search(r'typedef struct fami_route_t \{(.*?)\} fami_route_t;', src, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', ' ', body, flags=re.S)
    fields = [(m.group(2), ctypes.c_int if m.group(1) == 'int' else ctypes.c_long)
              for m in re.finditer(r'\b(int|long)\s+(\w+)\s*;', body)]
    return type('fami_route_t', (ctypes.Structure,), {'_fields_': fields})


Route = _route_struct()


class _Lib:
    def __init__(self):
        if not os.path.isfile(LIB_PATH):
            raise FamiError(
                "libfami_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C fami-pose_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.ncalls = 0                     # entry-point calls so far (a graph capture uses it to detect empty segments)
        import threading
        self._tls = threading.local()       # .bound: the fami_route_t THIS thread has bound (absent / None: process default)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        # FAMI_F32_SPLIT=0: f32 3x3 convolutions (forward, input and weight gradient) on the exact-f32 MFMA
        # (v_mfma_f32_16x16x4_f32) instead of the split-product kernels on the bf16 matrix pipe (conv_t4.hip S3,
        # conv_wgs3.hip; same accuracy class)
        # stored as the library's default state: fami_tune_reset / fami_conv_tune_lds(-1) restore it
        self.cdll.fami_tune_defaults(1 if options.flag('FAMI_F32_SPLIT') else 0)
        # The ablation switches compute WRONG results on purpose (upper-bound experiments of tools/): options.get() refuses them at
        # every read unless FAMI_ALLOW_WRONG=1 says the caller knows; here only the warning.
        wrong = [v for v in options.WRONG if options.flag(v)]
        if wrong:
            import sys
            print('[fami] WARNING: %s active -- results are WRONG by design (ablation run)' % ', '.join(wrong), file=sys.stderr, flush=True)
        self._env_route_switches()         # ... into the process default route (nothing is bound yet)

    def _env_route_switches(self):
        """The A/B switches of the environment that are ROUTE fields (options.py, 'library load'), written into the route the
        calling thread has bound: the process default at load, and every route new_route() hands out -- an Engine with a route
        of its own measures the same kernels as one on the process default."""
        if not options.flag('FAMI_T5'):      # A/B: the round-3 band kernel instead of the persistent one (conv_t5.hip)
            self.cdll.fami_conv_tune_lds(7000)
        for env, base in (('FAMI_WG16_TARGET', 21000), ('FAMI_WGS3_TARGET', 31000)):     # A/B: workgroup target of the weight-gradient kernels
            if options.get(env):
                self.cdll.fami_conv_tune_wgrad_lds(base + options.number(env))
        if options.get('FAMI_T5_ABL'):              # upper-bound experiment (WRONG results): only n chunks per convolution
            self.cdll.fami_conv_tune_lds(7600)
            self.cdll.fami_conv_tune_lds(7401)
            self.cdll.fami_conv_tune_lds(7700 + options.number('FAMI_T5_ABL'))
        if options.get('FAMI_T5_WG'):               # A/B: workgroups of the persistent grid / 8 (99: one job per workgroup)
            self.cdll.fami_conv_tune_lds(7500 + options.number('FAMI_T5_WG'))

    def new_route(self):
        """-> a fami_route_t with the library defaults (include/fami_route.h): the kernel-routing state an Engine owns."""
        r = Route()
        if self.cdll.fami_route_init(ctypes.byref(r)) != 0 or r.size != ctypes.sizeof(Route) or \
                self.cdll.fami_route_size() != ctypes.sizeof(Route):
            raise FamiError('fami_route_t layout differs between include/fami_route.h and libfami_hip.so -- rebuild the library')
        prev = getattr(self._tls, 'bound', None)
        self.bind(r)
        try:
            self._env_route_switches()
        finally:
            self.bind(prev)
        return r

    def bind(self, route):
        """Entry points called from this thread route by `route` (None: the process default) until the next bind.  The binding is
        thread-local on both sides; the bound object is kept alive here."""
        if route is not getattr(self._tls, 'bound', None):
            rc = self.cdll.fami_route_bind(None if route is None else ctypes.byref(route))
            if rc != 0:
                raise FamiError('fami_route_bind failed (%d): %s' % (rc, self.cdll.fami_last_error().decode()))
            self._tls.bound = route

    def call_routed(self, route, name, *args):
        """call() under `route` (an Engine's): bound first, left bound."""
        self.bind(route)
        self.ncalls += 1
        rc = getattr(self.cdll, name)(*args)
        if self.protos[name][0] is ctypes.c_int and rc != 0:
            raise FamiError('%s failed (%d): %s' % (name, rc, self.cdll.fami_last_error().decode()))
        return rc

    def call(self, name, *args):
        """An entry point under the process default route (callers with a route of their own: call_routed)."""
        self.bind(None)
        self.ncalls += 1
        rc = getattr(self.cdll, name)(*args)
        if self.protos[name][0] is ctypes.c_int and rc != 0:
            raise FamiError('%s failed (%d): %s' % (name, rc, self.cdll.fami_last_error().decode()))
        return rc


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def loaded():
    return _lib is not None

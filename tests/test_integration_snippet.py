"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would paste into flow.py.  It sets no
argtypes, so a missing argument silently shifts the stream handle into a data pointer: execute the snippet against a
recording stand-in for the library and check every call's arity and pointer/integer kinds against _lib.SIGNATURES
(which tests/test_cabi_exports.py ties to include/gpode.h and to the built .so)."""
import ctypes
import os
import re
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeLib:
    def __init__(self, signatures):
        self.sig, self.calls = signatures, []

    def __getattr__(self, name):
        if name not in self.sig:
            raise AttributeError(name)
        res, argt = self.sig[name]

        def fn(*args):
            assert len(args) == len(argt), '%s: %d arguments passed, the header takes %d' % (name, len(args), len(argt))
            for i, (a, t) in enumerate(zip(args, argt)):
                if t is ctypes.c_int:
                    assert isinstance(a, int) and not isinstance(a, bool), '%s arg %d: expected an int, got %r' % (name, i, a)
                elif t is ctypes.c_void_p:
                    assert isinstance(a, ctypes.c_void_p), '%s arg %d: expected a pointer, got %r' % (name, i, a)
                else:                                   # size_t* outputs arrive as byref(...)
                    a._obj.value = 64
            self.calls.append(name)
            return b'' if res is ctypes.c_char_p else 0
        return fn


def test_section_b_snippet_matches_the_header():
    from vae_gp_ode_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = [b for b in blocks if 'hip_flow_forward' in b]
    assert len(code) == 1
    fake = _FakeLib(_lib.SIGNATURES)
    real_cdll = ctypes.CDLL
    ctypes.CDLL = lambda *_a, **_k: fake
    try:
        ns = {}
        exec(code[0], ns)
    finally:
        ctypes.CDLL = real_cdll
    ns['_st'] = lambda: ctypes.c_void_p(0)              # no GPU here: the stream handle is just a pointer-sized value
    Di = Do = 2
    M, S, N, T = 4, 8, 3, 5
    P = lambda *s: types.SimpleNamespace(optvar=torch.zeros(*s))
    layer = types.SimpleNamespace(kernel_n='RBF', D_in=Di, D_out=Do, M=M, S=S, inducing_loc=P(M, Di), Um=P(M, Do),
                                  Us_sqrt=P(Do, M * (M + 1) // 2),
                                  kern=types.SimpleNamespace(unconstrained_lengthscales=torch.zeros(Do, Di),
                                                             unconstrained_variance=torch.zeros(Do)))
    noise = dict(eps_u=torch.zeros(M, Do), rff_w=torch.zeros(S, Do), rff_eps=torch.zeros(Di, S, Do), rff_u=torch.zeros(1, S, Do))
    zt = ns['hip_flow_forward'](layer, torch.zeros(N, Di), torch.zeros(T), 1, 'rk4', noise)
    assert tuple(zt.shape) == (N, T, Di)
    assert fake.calls == ['gpode_cache_sizes', 'gpode_cache_build_fwd', 'gpode_rollout_fwd']

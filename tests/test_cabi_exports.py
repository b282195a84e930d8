"""CPU-only: the C-ABI library loads and exports every symbol include/gpode.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'gpode.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gpode_\w+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from vae_gp_ode_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n


def test_binding_table_matches_header():
    from vae_gp_ode_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_host_only_entry_points():
    """Entry points that do not touch the GPU: version string, supported-dims table, size queries, errors."""
    from vae_gp_ode_amd import _lib, ops
    lib = _lib.load()
    assert b'gfx950' in lib.gpode_version()
    assert lib.gpode_supported(0, 6, 6) == 1 and lib.gpode_supported(0, 6, 3) == 1 and lib.gpode_supported(1, 6, 6) == 1
    assert lib.gpode_supported(1, 6, 3) == 0 and lib.gpode_supported(0, 7, 5) == 0
    pf, wf = ops.cache_sizes('RBF', 6, 6, 100, 256)
    # 4 feature groups x 6 dims x 2 quads x 64 lanes + 2 inducing groups x 3 quads x 64 lanes (float4) + 36 uniforms
    assert pf == 4 * (4 * 6 * 2 * 64 + 2 * 3 * 64) + 36
    assert wf > 6 * 128 * 128
    try:
        ops.cache_sizes('RBF', 7, 5, 10, 10)
    except _lib.GpodeError as e:
        assert 'no specialisation' in str(e)
    else:
        raise AssertionError('expected GpodeError')


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under vae-gp-ode_amd/ may import it."""
    pkg = os.path.join(ROOT, 'vae-gp-ode_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dp, f)

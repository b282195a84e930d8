"""Determinism of the graph-replayed training loop at tiny batch sizes (1 rank; batches of 2, 2, 1 sequences)."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(tmp, tag, extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'vae_gp_ode_amd.main', '--task', 'synthetic', '--Ndata', '5', '--Ntest', '4', '--batch', '2', '--T', '6', '--solver', 'rk4',
           '--num_inducing', '16', '--num_features', '32', '--lr', '1e-3', '--log_freq', '1', '--Nepoch', '4', '--save', 'results/' + tag] + extra
    r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    log = glob.glob(os.path.join(tmp, 'results', tag + '_*', 'logs'))[0]
    return [float(m.group(1)) for m in re.finditer(r'elbo\s+(-?[\d.]+)\(', open(log).read())]
tmp = tempfile.mkdtemp()
runs = {}
for tag, extra in (('e', ['--device_noise', 'True']), ('g_a', ['--hip_graph', 'True']), ('g_b', ['--hip_graph', 'True']), ('g_c', ['--hip_graph', 'True'])):
    runs[tag] = run(tmp, tag, extra)
n = min(len(v) for v in runs.values())
print('iter   ' + '   '.join('%12s' % k for k in runs))
for i in range(n):
    print('%4d   ' % i + '   '.join('%12.2f' % runs[k][i] for k in runs))

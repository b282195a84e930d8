"""Where do the eager and the graph-replayed training runs part?  Runs main.py four ways (1 / 2 ranks x eager / graph) on the
same synthetic data and device noise and prints the per-iteration elbo lines side by side."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run(tmp, tag, world, extra, port):
    env = dict(os.environ, GPODE_DIST_BACKEND='gloo', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    base = ['-m', 'vae_gp_ode_amd.main', '--task', 'synthetic', '--Ndata', '10', '--Ntest', '4', '--batch', '4', '--T', '6', '--solver', 'rk4',
            '--num_inducing', '16', '--num_features', '32', '--lr', '1e-3', '--log_freq', '1', '--Nepoch', '6', '--save', 'results/' + tag,
            '--sync_bn', 'False'] + extra + os.environ.get('DBG_EXTRA', '').split()
    if world == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', str(port)] + base
    r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    log = glob.glob(os.path.join(tmp, 'results', tag + '_*', 'logs'))[0]
    return [float(m.group(1)) for m in re.finditer(r'elbo\s+(-?[\d.]+)\(', open(log).read())]

tmp = tempfile.mkdtemp()
runs = {}
for tag, world, extra, port in (('e2', 2, ['--device_noise', 'True'], 29571), ('e2o', 2, ['--device_noise', 'True', '--gp_side_stream', 'True'], 29573),
                                ('g2a', 2, ['--hip_graph', 'True'], 29572), ('g2b', 2, ['--hip_graph', 'True'], 29574)):
    runs[tag] = run(tmp, tag, world, extra, port)
n = min(len(v) for v in runs.values())
print('iter   ' + '   '.join('%12s' % k for k in runs))
for i in range(n):
    print('%4d   ' % i + '   '.join('%12.2f' % runs[k][i] for k in runs))

"""CPU-only: the data path (vae_gp_ode_amd/data) against fixtures captured from the reference's data/utils.py and
data/mnist.py (tests/golden/make_golden.py::data_case) and against a synthetic rot-mnist.mat."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'data_path.npz'))


def test_dataset_items_match_reference(g):
    from vae_gp_ode_amd.data.utils import Dataset
    ds = Dataset(g['seqs'])
    assert len(ds) == 3
    for i in range(3):
        assert ds[i].shape == (16, 1, 28, 28) and ds[i].dtype == torch.float32
        assert np.array_equal(ds[i].numpy(), g['items'][i])         # same fp32 operations: bit-exact


def test_rot_start_matches_reference(g):
    from vae_gp_ode_amd.data.mnist import rot_start
    np.random.seed(int(g['rot_seed']))
    out = rot_start(torch.tensor(g['seqs']).view(3, 16, 1, 28, 28), 16, 3)
    assert np.array_equal(out.numpy(), g['rot'])


def test_rotate_img_matches_reference(g):
    from vae_gp_ode_amd.data.mnist import rotate_img
    out = rotate_img(g['digits'], g['angles'])
    assert out.shape == (3, 8, 28, 28) and out.dtype == g['rotated'].dtype
    assert np.array_equal(out, g['rotated'])
    assert np.array_equal(out[:, 0], g['digits'])


def test_labelled_dataset():
    from vae_gp_ode_amd.data.utils import Dataset_labels
    x, y = np.arange(12.).reshape(6, 2), np.arange(6).reshape(3, 2)
    ds = Dataset_labels(x, y)
    assert len(ds) == 6 and ds[4][1] == 4 and np.array_equal(ds[4][0], x[4])


def _write_mat(root, n_per_digit=14, seed=5):
    import scipy.io as sio
    rng = np.random.RandomState(seed)
    Y = np.tile(np.array([3, 7]), n_per_digit)
    X = (rng.randint(0, 256, (Y.size, 16, 784)) / 255).astype(np.float32)
    os.makedirs(os.path.join(root, 'rot_mnist'), exist_ok=True)
    sio.savemat(os.path.join(root, 'rot_mnist', 'rot-mnist.mat'), {'X': X[None], 'Y': Y[None]})     # squeezed by the loader
    return X, Y


def _args(root, **kw):
    a = dict(data_root=str(root), task='mnist', mask=True, value=3, Ndata=8, Ntest=3, batch=4, T=16, device='cpu', seed=1,
             save=str(root))
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_load_mnist_data_split_and_normalisation(tmp_path):
    from vae_gp_ode_amd.data.wrappers import load_data
    X, Y = _write_mat(tmp_path)
    threes = X[Y == 3]
    trainset, testset = load_data(_args(tmp_path), plot=False)
    assert len(trainset) == 2 and len(testset) == 1
    tr = torch.cat(list(trainset), 0)
    want = (torch.tensor(threes[:8]).view(8, 16, 1, 28, 28) - 0.1307) / 0.3081
    # shuffled: match rows by content
    order = [int(torch.where((want.flatten(1) == row.flatten()).all(1))[0]) for row in tr]
    assert sorted(order) == list(range(8))
    te = torch.cat(list(testset), 0)
    want_te = (torch.tensor(threes[11:14]).view(3, 16, 1, 28, 28) - 0.1307) / 0.3081     # after the 3-long validation block
    assert sorted(map(float, te.flatten(1).sum(1))) == pytest.approx(sorted(map(float, want_te.flatten(1).sum(1))))


def test_load_mnist_data_errors(tmp_path):
    from vae_gp_ode_amd.data.wrappers import load_data
    with pytest.raises(FileNotFoundError):
        load_data(_args(tmp_path), plot=False)
    with pytest.raises(ValueError):
        load_data(_args(tmp_path, task='other'), plot=False)
    _write_mat(tmp_path, n_per_digit=5)
    with pytest.raises(ValueError):
        load_data(_args(tmp_path), plot=False)               # 5 sequences < 8 + 2 * 3


def test_frame_loaders(tmp_path):
    from vae_gp_ode_amd.data.mnist import load_mat_mnist_data, load_rotating_mnist_data
    X, Y = _write_mat(tmp_path)
    tr, te = load_mat_mnist_data(_args(tmp_path, batch=16), plot=False)
    assert len(tr.dataset) == 8 * 16 and len(te.dataset) == 3 * 16
    x, t = tr.dataset[17]
    assert x.shape == (1, 28, 28) and int(t) == 1 and np.array_equal(x.numpy().ravel(), X[Y == 3][1, 1])
    arr = np.random.RandomState(0).rand(4, 8, 28, 28).astype(np.float32)
    np.save(tmp_path / 'rot.npy', arr)
    loader = load_rotating_mnist_data(str(tmp_path / 'rot.npy'), _args(tmp_path, n_angle=8, batch=5), plot=False)
    assert len(loader.dataset) == 32 and int(loader.dataset[13][1]) == 5


def test_create_rotating_dataset_from_arrays(tmp_path):
    from vae_gp_ode_amd.data.mnist import create_rotating_dataset, rotate_img
    rng = np.random.RandomState(3)
    imgs, labs = (rng.rand(20, 28, 28) * 255).astype(np.uint8), np.tile(np.array([3, 5]), 10)
    np.random.seed(11)
    tr, te = create_rotating_dataset(str(tmp_path), digit=3, train_n=4, test_n=2, n_angles=6, images=imgs, labels=labs)
    assert tr.shape == (4, 6, 28, 28) and te.shape == (2, 6, 28, 28) and tr.dtype == np.float32
    np.random.seed(11)
    pick = imgs[labs == 3][np.random.randint(0, 10, 4)]
    assert np.array_equal(tr, (rotate_img(pick, np.rad2deg(np.linspace(0, 2 * np.pi, 6)[1:])) / 255).astype(np.float32))
    with pytest.raises(FileNotFoundError):
        create_rotating_dataset(str(tmp_path))


def test_resident_loader_epochs():
    from vae_gp_ode_amd.data.utils import ResidentLoader
    items = torch.arange(10.)[:, None].repeat(1, 3)
    ld = ResidentLoader(items, 4, shuffle=True, device='cpu', seed=9)
    assert len(ld) == 3
    e1, e2 = [b.clone() for b in ld], [b.clone() for b in ld]
    assert [b.shape[0] for b in e1] == [4, 4, 2]
    for ep in (e1, e2):
        assert sorted(torch.cat(ep)[:, 0].tolist()) == list(range(10))
    assert not torch.equal(torch.cat(e1), torch.cat(e2))                 # a fresh permutation per epoch
    again = [b.clone() for b in ResidentLoader(items, 4, shuffle=True, device='cpu', seed=9)]
    assert torch.equal(torch.cat(again), torch.cat(e1))                  # seeded
    plain = list(ResidentLoader(items, 4, shuffle=False, device='cpu'))
    assert torch.equal(torch.cat(plain), items)

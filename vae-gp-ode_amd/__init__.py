"""MI355X-native drop-in for the VAE-GP-ODE hot path (GP-ODE integrator + conv VAE + ELBO).

Layout
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/gpode.h)
  _lib.py          ctypes binding of libgpode_hip.so (fails loudly when the library is missing)
  ops.py           tensor-level wrappers (raw device pointers + current HIP stream)
  model/           host-side mirror of the reference's experiments/model/ operator API
  main.py          experiments/main.py argument surface + training loop
"""
__version__ = '0.1.0'

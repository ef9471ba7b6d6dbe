"""CPU oracle for the bhnerf hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain NumPy (``oracle_np``) and in PyTorch-CPU
(``oracle_torch``, for autograd/Adam and for the timed CPU baseline), the
arithmetic of the one hot path of aviadlevis/bhnerf: velocity warp -> positional
encoding -> skip-MLP -> sigmoid/masks -> radiative-transfer ray sum -> chi^2 loss
-> Adam.  Every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The shipped package ``bhnerf_amd``
never imports ``oracle``; its device path fails loudly if the HIP library is
missing.

Parity pinning: the importable pieces (warp, posenc, fill, radiative transfer,
image_plane_prediction, loss_fn_image, loss_fn_eht) are checked against golden
vectors produced by running the reference's own NumPy code path in the build
container (``tests/golden/make_golden.py``).  The pieces that cannot be imported
without JAX/flax/optax (MLP.__call__, the NeRF_Predictor glue, he_uniform init,
optax.adam + polynomial_schedule) are restated from network.py:49-62, 219-233,
173-174 and the libraries' published formulas: PARITY UNPINNED for those three
(flax Dense, optax adam, jax PRNG) -- see DESIGN.md.
"""

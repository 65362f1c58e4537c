"""CPU oracle for the GAN-DANet G+D training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product package
(``gan-danet_amd/``) never imports from here and has no CPU fallback.

What it is: a PyTorch-CPU fp32/fp64 restatement of the reference algorithm
(reference = /root/reference, Aster32/GAN-DANet):

* ``functional.py`` -- the arithmetic of every op on the path written out
  explicitly (attention, batch-norm, losses, AdamW ...), each function citing
  the reference ``file:line`` it follows.
* ``modules.py``    -- ``nn.Module`` containers with the reference class names,
  constructor signatures and ``state_dict`` keys, built on ``functional``.
* ``step.py``       -- the G+D update of ``ModelTrainer.train``
  (GAN_DANet_train.ipynb:L225-272) as a plain function.

Where the arithmetic really lives: the reference calls third-party PyTorch
ATen kernels (conv2d, batch_norm, bmm, softmax, upsample_*; unpinned,
requirement.yml:15-17) and torchvision's VGG19 topology (losses.py:10).  The
oracle therefore runs on this image's torch 2.10 CPU kernels.

Pinning status: the reference ships NO tests, golden vectors or fixtures for
this path.  The oracle is pinned instead against outputs
of the reference itself, imported by file path in the build container
(generating scripts ``tests/golden/make_golden.py``, ``make_golden_a14.py``, ``make_golden_data.py``; the fixtures
they wrote sit next to them and ``tests/test_oracle_golden.py`` checks every oracle function against them).  ``PerceptualLoss`` cannot be constructed from the
reference here (needs torchvision, absent): that one term is
"parity unpinned" -- restated from losses.py:13-73 plus the published VGG19
"E" feature stack, and checked oracle-vs-HIP only.
"""

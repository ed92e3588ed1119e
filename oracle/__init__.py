"""CPU oracle for the avddpg hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy restatement of the reference algorithm
(cboin1996/avddpg) for the path named by BASELINE.json: platoon dynamics step,
reset, OU noise, replay ring/sample, actor/critic MLP forward/backward
(`Trainer.learn`), TF-formulation Adam, Polyak target update and the federated
mean.  Every function cites the reference file:line it follows.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / reported baseline only.
Nothing under ``avddpg_amd/`` imports it; the product path fails loudly when the
HIP extension is missing.

Pinning status
--------------
* environment / reset / OU noise / replay buffer / trainer inner-loop RNG
  interleaving / evaluator rollout (decentralized and centralized): PINNED against
  golden vectors captured by importing the reference's own NumPy code in the build
  container (``tests/golden/make_golden.py`` -> ``tests/golden/g1..g6, g8, g9``),
  checked by ``tests/test_oracle_golden.py``.
* Dense / BatchNormalization / tanh / GradientTape / Adam / federated mean:
  **parity unpinned**.  That arithmetic lives in third-party
  ``tensorflow==2.4.1`` (reference ``requirements.txt:2``) which is absent from
  the image and from ``/root/reference``; the reference holds no test or golden
  vector for it.  The restatement follows the reference call sites
  (``agent/model.py``, ``workers/trainer.py:472-508``, ``agent/ddpgagent.py:31-55``,
  ``src/server/federated.py``) plus TF 2.4.1's published semantics (Keras
  BatchNormalization inference form with eps=1e-3; ``ApplyAdam`` functor of
  ``tensorflow/core/kernels/training_ops.cc``), and is cross-checked against
  torch-CPU float64 autograd and finite differences in
  ``tests/test_oracle_mlp.py``; the only reference-derived known answer is the
  hand-computable table of ``src/server/test_federated.py:26-42`` (g7).
"""

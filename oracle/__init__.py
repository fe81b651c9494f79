"""CPU oracle for the A4NT adversarial-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The
product path (``a4nt-author-masking_amd``) never routes through this package.

It is a pure-PyTorch fp32 restatement (CPU, Python 3) of the reference's four
model files and of the adversarial step of ``train_adversarial_translator.py``.
Pinned: ``tests/golden/*.npz`` hold inputs/outputs produced by importing the
reference's own ``models/*.py`` in the build container (``oracle/gen_golden.py``
is the generating script); ``tests/test_oracle_golden.py`` checks this
restatement against them.
"""
from .models import (NoiseSource, CharTranslator, CharLstm, CharCNN, get_classifier,  # noqa: F401
                     gumbel_softmax_sample, sample_gumbel)
from .adv_step import (adv_forward_pass, d_update, g_update, AdvState, make_optimizers,  # noqa: F401
                       eval_forward_pass)

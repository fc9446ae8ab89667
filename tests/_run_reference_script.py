"""Test launcher (never part of the product): runs an UNMODIFIED script of the reference (e.g.
/root/reference/scripts/extract.py) against this repo's ``esm`` package on a machine without a GPU.

The engine has no CPU path by design, so ONLY inside this launcher ``ESM2.forward`` is replaced by the oracle
(test infrastructure, oracle/esm2_oracle.py).  What the test then exercises is everything else the script touches:
``esm.pretrained.load_model_and_alphabet`` on a checkpoint file, ``FastaBatchedDataset``, the batch converter under
a torch DataLoader, ``isinstance(model, MSATransformer)``, ``model.eval()`` / ``num_layers``, the keys and shapes of
the output dict, the per-sequence result files.

    python tests/_run_reference_script.py /root/reference/scripts/extract.py <script arguments...>
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import esm  # noqa: E402  (this repo's drop-in package)
import esm_amd.esm2  # noqa: E402
from oracle.esm2_oracle import esm2_forward  # noqa: E402

assert esm.__file__.startswith(ROOT), esm.__file__


def _oracle_forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False):
    sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
    return esm2_forward(sd, tokens.cpu(), self.num_layers, self.attention_heads, repr_layers=repr_layers,
                        need_head_weights=need_head_weights, return_contacts=return_contacts)


esm_amd.esm2.ESM2.forward = _oracle_forward
script = sys.argv[1]
sys.argv = [script] + sys.argv[2:]
runpy.run_path(script, run_name="__main__")

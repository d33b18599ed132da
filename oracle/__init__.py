"""CPU oracle for the STARCOP segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product package
(``starcop_amd``) never imports this package and fails loudly when its HIP
library is missing.

Parity pin status (see DESIGN.md section "Oracle"):
  * mag1c / normaliser / thresholds / metrics / padding / band-ratio
    restatements are PINNED against the reference itself, imported in the build
    container (tests/golden/make_golden.py wrote tests/golden/*.npz).
  * the U-Net restatement (oracle/unet_ref.py) follows the third-party
    ``segmentation_models_pytorch.Unet('mobilenet_v2')`` which is absent from
    /root/reference (unpinned dependency, requirements.txt:9): its *structure*
    is pinned by the reference's own Lightning log (6 629 233 parameters,
    notebooks/(bonus)_training_demo.ipynb cell 19) and by state_dict key names;
    its bytes are "parity unpinned".
"""

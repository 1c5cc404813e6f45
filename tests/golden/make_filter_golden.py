#!/usr/bin/env python3
"""Generates tests/golden/reference_filter_golden.json by running the REFERENCE's own baseline-classifier module
(fgvc/models/cal.py: WSDAN_CAL, eval mode; fgvc/models/resnet.py features) in the build container on seeded inputs, with the
state dict this repo's synthetic-weight generator produces for the same architecture (the reference model loads it through
its own load_state_dict).  Only the seeds and the resulting logits are committed: the GPU box regenerates the identical
weights / input from the seeds and compares the oracle restatement and the HIP launch graph against these numbers.

    python tests/golden/make_filter_golden.py"""
import json
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))
REF = "/root/reference"
OUT = HERE / "reference_filter_golden.json"


def main():
    import make_golden as MG
    sys.path.insert(0, REF)
    MG.stub_modules()
    import fgvc.models.cal as cal
    import saspa_aug_amd  # noqa: F401
    from saspa_aug_amd import config as CFG
    from saspa_aug_amd import weights as W
    golden = {}
    for tag, net, base, ncls, res in (("resnet50", "resnet50", CFG.WSDAN_CAL_R50, 12, 224), ("resnet101", "resnet101", CFG.WSDAN_CAL_R101, 30, 224)):
        cfg = dict(base, num_classes=ncls)
        sd = W.synth_state_dict("cal", cfg, 21)
        model = cal.WSDAN_CAL(num_classes=ncls, net=net, print_func=lambda *a: None)
        missing = [k for k in model.state_dict() if k not in sd and not k.endswith("num_batches_tracked")]
        extra = [k for k in sd if k not in model.state_dict()]
        assert not missing and not extra, (missing[:5], extra[:5])
        model.load_state_dict(sd)
        model.eval()
        x = torch.randn((2, 3, res, res), generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            p = model(x)[0]
        golden[tag] = dict(cfg=dict(cfg, layers=list(cfg["layers"])), weight_seed=21, input_seed=5, input_shape=list(x.shape),
                           logits=p.double().tolist())
    json.dump(golden, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()

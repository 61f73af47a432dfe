#!/usr/bin/env python3
"""Golden for the post-processing caller of detect(): the reference's ``_dict_from_results``
(/root/reference/pytorch_yolo/utils/utils.py:306-327) run on tests/_cases.results_case().

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_results.py      # needs /root/reference (build container only)
"""
import json
import os

import torch

from make_golden import HERE, import_reference      # sets sys.path, stubs the reference's parent package


def main():
    import _cases as C
    import_reference()
    from pytorch_yolo.utils.utils import _dict_from_results
    dets, paths, shapes, cur = C.results_case()
    targets = [None if d is None else torch.from_numpy(d.copy()) for d in dets]
    out = _dict_from_results({}, targets, paths, shapes, cur)
    with open(os.path.join(HERE, "dict_from_results.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

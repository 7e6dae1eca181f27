#!/usr/bin/env python3
"""One case of tools/fuzz_options.py and variants of it with single options changed (to find
which option a disagreement with the oracle hangs on):
python tools/fuzz_case.py <seed> [key=value ...] [-- key=value ...] ..."""
import ast
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("fuzz_options",
                                              os.path.join(ROOT, "tools", "fuzz_options.py"))
fo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fo)


def main():
    seed = int(sys.argv[1])
    variants, cur = [], {}
    for a in sys.argv[2:]:
        if a == "--":
            variants.append(cur); cur = {}
        else:
            k, v = a.split("=", 1)
            cur[k] = ast.literal_eval(v)
    variants.append(cur)
    device = torch.device("cuda:0")
    for over in variants:
        c = fo.case(seed)
        c.update(over)
        try:
            problems = fo.run(c, seed, device)
        except Exception as error:
            problems = ["exception: {!r}".format(error)]
        print(over or "(as drawn)", "->", len(problems), "problems")
        for p in problems[:4]:
            print("     ", p)


if __name__ == "__main__":
    main()

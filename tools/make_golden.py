#!/usr/bin/env python
"""Writes the golden fixtures under tests/golden/ from the CPU oracle (oracle/qg_oracle.c, oracle/qgnode_oracle.c).

    python tools/make_golden.py            # (re)generate every file
    python tools/make_golden.py NAME ...   # only the named cases

Each `<case>.npz` holds the inputs of the case (`in_*`) and every result the case function of tests/golden_cases.py
returns, fp64, uncompressed-exact.  `p0_32x3.bas` is the fixed float32 restart file of the 10-step double-gyre run
(layout of msqg/auxiliar_input.h:128-140).  Provenance: the reference cannot be built here (Basilisk-C, needs qcc) and
holds no vectors for this path, so these files are outputs of the restatement -- they pin the oracle and the kernels
between rounds, they do not prove parity with the reference.  Regenerating must reproduce the
committed files bit for bit (tests/test_golden_oracle.py checks exactly that); a deliberate change of the oracle has to
come with regenerated files in the same commit.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "4")
import golden_cases as gc  # noqa: E402
import orc  # noqa: E402


def write_p0bas():
    """psi of the synthetic IC at 32^2 x 3, rounded to float32 by the .bas format itself"""
    o = orc.Oracle(gc.wl.double_gyre_params(32, 3), quiet=1)
    o.set(orc.PSI, gc.wl.synthetic_psi(3, 32, 32))
    assert o.write_bas(orc.PSI, gc.P0BAS) == 0


def main(names):
    os.makedirs(gc.GOLDEN, exist_ok=True)
    if not names or not os.path.exists(gc.P0BAS):
        write_p0bas()
    for table, make in ((gc.CASES, gc.OracleModel), (gc.LEX_CASES, gc.OracleModel), (gc.NODE_CASES, gc.NodeOracleModel)):
        for name, (fn, kw, builder) in table.items():
            if names and name not in names:
                continue
            inp = builder() if builder else {}
            kw = dict(kw)
            smoother = kw.pop("smoother", None)
            mk = (lambda txt, **o: make(txt, smoother=smoother, **o)) if smoother is not None else make
            out = fn(mk, inp, **kw)
            assert not (set(out) & set(inp))
            path = os.path.join(gc.GOLDEN, name + ".npz")
            np.savez_compressed(path, **inp, **{k: np.asarray(v, dtype=np.float64) for k, v in out.items()})
            print(f"{name}: {len(inp)} inputs, {len(out)} results, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1:])

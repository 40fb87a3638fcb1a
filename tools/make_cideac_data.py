#!/usr/bin/env python3
"""Extracts the tabulated sea-ice scattering attenuation coefficients (ln of the attenuation per floe, Kohout & Meylan
2008 Fig. 6 as received by ECMWF, thickness 0.2 .. 3.7 m x period 6 .. 16 s) that cigetdeac.F90:85-552 assigns to
CIDEAC(6:16, 1:36) and writes them as plain numbers to ecwam_amd/data/cideac_kohout_meylan.txt (36 rows = ice thickness
HICMIN + (IH-1)*DHIC, 11 columns = wave period 6..16 s).  Run once in the build container (the reference is not present on
the GPU box); the rest of the table (periods 1..5 s) is filled in by ecwam_amd/tables.py with cigetdeac.F90:76-82,553-559.
"""
import os
import re

import numpy as np

SRC = "/root/reference/src/ecwam/cigetdeac.F90"
HERE = os.path.dirname(os.path.abspath(__file__))
# the product's copy and the oracle's own (the checker reads nothing of the product)
OUTS = [os.path.join(HERE, "..", "ecwam_amd", "data", "cideac_kohout_meylan.txt"), os.path.join(HERE, "..", "oracle", "data", "cideac_kohout_meylan.txt")]

t = np.full((36, 16), np.nan)
for line in open(SRC):
    m = re.match(r"\s*CIDEAC\(\s*(\d+)\s*,\s*(\d+)\s*\)\s*=\s*(-?[0-9.]+)_JWRB", line)
    if m:
        t[int(m.group(2)) - 1, int(m.group(1)) - 1] = float(m.group(3))
blk = t[:, 5:]
assert not np.isnan(blk).any(), "table block incomplete"
assert t[0, 0] == -2.0
for OUT in OUTS:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write("# ln(attenuation coefficient per floe), rows: ice thickness 0.2 + 0.1*i m (i = 0..35), columns: wave period 6..16 s\n")
        f.write("# data of Kohout & Meylan (2008) as tabulated in ecWAM 1.5.13 cigetdeac.F90:85-552 (extracted by tools/make_cideac_data.py)\n")
        for r in blk:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    print("wrote", OUT, blk.shape)

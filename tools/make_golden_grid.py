"""Generates tests/golden/grid_description_O*.txt by RUNNING the reference's own grid script (share/ecwam/scripts/ecwam_grids.py,
a Python CLI) in the development container.  The fixtures are its output (data: resolution, first/last latitude, west/east,
iper, irgg, ny, the ny row lengths), not its source.  Only usable where /root/reference exists."""
import os
import subprocess
import sys

REF = "/root/reference/share/ecwam/scripts/ecwam_grids.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for g in ("O48", "O320", "O640"):
    txt = subprocess.run([sys.executable, REF, g, "--grid_description"], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(OUT, f"grid_description_{g}.txt"), "w") as f:
        f.write(txt)
    print(g, len(txt.splitlines()), "lines")

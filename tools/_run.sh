python tools/aten_sites.py 2>&1 | tail -45

"""register / scratch / occupancy table from a `hipcc -S` listing (the .set NAME.num_vgpr ... symbols).  usage: isa_resources.py FILE.s [SUBSTRING]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = {}
for m in re.finditer(r'\.set (\S+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)', s):
    rows.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
for k, v in rows.items():
    if sub in k and 'num_vgpr' in v:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem); dem = re.sub(r"^void ", "", dem); dem = re.sub(r"\(.*", "", dem)
        print(f"{dem[:70]:70s} vgpr {v.get('num_vgpr'):4d} agpr {v.get('num_agpr', 0):4d} scratch {v.get('private_seg_size', 0):5d}")

"""Stamp the profile set of a round with the commit it was measured at (run locally after copying gpurun_out/profiles/<tag>_*
into profiles/): writes profiles/<tag>_COMMIT.txt and adds "profile_commit" to every <tag>_*.json that holds one JSON
object.  usage: stamp_profiles.py r03"""
import glob
import json
import subprocess
import sys

tag = sys.argv[1]
head = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
dirty = subprocess.check_output(["git", "status", "--porcelain", "--", "lion_amd", "bench.py", "oracle", "include"], text=True).strip()
note = head + (" + uncommitted changes:\n" + dirty if dirty else " (lion_amd/, bench.py, oracle/, include/ clean)")
open(f"profiles/{tag}_COMMIT.txt", "w").write(
    f"profiles/{tag}_* were measured on the tree at commit {note}\n(one gpurun call, tools/collect_profiles.sh {tag}; the GPU box carries no .git, the stamp is applied here)\n")
for f in glob.glob(f"profiles/{tag}_*.json"):
    try:
        d = json.loads(open(f).read())
    except Exception:
        continue
    if isinstance(d, dict):
        d["profile_commit"] = head
        open(f, "w").write(json.dumps(d) + "\n")
print(note)

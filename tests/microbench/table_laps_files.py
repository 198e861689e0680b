import os, sys
sys.path[:0]=["/root/repo","/root/repo/tests"]
import arpeggia_amd as aa
aa.debug_set("timing", 1)
ctx = aa.Context(0)
for name in ("1ubq","6bft"):
    s = aa.load_model(f"/root/repo/tests/data/{name}.pdb")
    for i in range(4):
        print(f"--- {name} call {i}", flush=True)
        t = ctx.get_contacts(s, "/", 0.1, 6.5)

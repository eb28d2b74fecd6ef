import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
order = sys.argv[1]
if order == "lib_first":
    ctx = pkg.default_context(); print("lib ctx ok")
import torch
print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
try:
    s = torch.cuda.Stream(); print("stream ok", s.cuda_stream)
except Exception as e:
    print("stream failed:", str(e)[:100])
if order != "lib_first":
    ctx = pkg.default_context(); print("lib ctx ok (after torch)")

"""The table of tuning switches of INTEGRATION.md section 3b, printed from the library's own registry
(gpet_option_info): python tools/options_table.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_process_edge_trace_amd import _lib  # noqa: E402

print("| option | default | range | environment | meaning |")
print("|---|---|---|---|---|")
for name, o in _lib.options().items():
    print("| `%s` | %d | %d … %d | `GPET_%s` | %s |" % (name, o["default"], o["lo"], o["hi"], name.upper(), o["doc"].replace("|", "/")))

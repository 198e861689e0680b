"""Contact tables from the HOST assembly (tests/hosttable/table_host.inl), which only the test-only library contains.  Run by
tests/test_gpu_parity.py::test_device_table_equals_the_host_assembly in a process of its own:
    ARPEGGIA_AMD_LIB=tests/hosttable/build/libarpeggia_amd_hosttable.so python tests/hosttable/dump_table.py
Prints one JSON object {case: {groups: [csv lines]}}."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
assert "hosttable" in os.environ.get("ARPEGGIA_AMD_LIB", "")
import arpeggia_amd as aa  # noqa: E402

aa.debug_set("table_host", 1)  # (only this test-only library contains the host assembly)
import synth  # noqa: E402
from test_gpu_parity import _table_lines, table_cases  # noqa: E402

ctx = aa.Context(0)
out = {}
for name, s in table_cases().items():
    out[name] = {g: _table_lines(ctx.get_contacts(s, g, 0.1, 6.5)) for g in ("/", "A,B/")}
print(json.dumps(out))

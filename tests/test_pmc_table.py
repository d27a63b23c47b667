"""profiles/pmc_traffic.json is a static table of rocprofv3 PMC measurements that bench.py reports `utilisation` and
`roofline.traffic` from.  Every entry records the code it was measured ON: the SHA-256 of the contact kernel's machine
code (shpair/codeobj.py), the ring-group size and the waves per pair.  This test (no GPU needed: it reads the built
library) fails when the shipped libshpair.so no longer contains the kernels the table was measured on — re-run
tools/pmc_refresh.sh on the GPU box and merge (tools/pmc_table.py --merge-json) before committing a kernel change."""
import json
import os

from shpair import capi, codeobj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table():
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    return {k: v for k, v in t.items() if isinstance(v, dict)}


def test_every_entry_names_the_code_it_was_measured_on():
    tab = _table()
    assert "100000:6:16:1:1.25:sharp:jpoly" in tab            # the headline workload
    assert not [k for k in tab if k.count(":") == 5 and k.endswith(":sharp")], "body-frame entries of kernels that no longer exist"
    for k, e in tab.items():
        for f in ("kernel_symbol", "kernel_hash", "ring_rows", "waves_per_pair", "commit", "files"):
            assert e.get(f) not in (None, ""), (k, f)
        assert os.path.exists(os.path.join(ROOT, e["files"])), e["files"]


def test_the_table_is_not_stale_at_head():
    hashes = codeobj.kernel_hashes(capi.library_path())
    for k, e in _table().items():
        assert e["kernel_symbol"] in hashes, (k, "the library has no such kernel any more")
        assert hashes[e["kernel_symbol"]] == e["kernel_hash"], \
            (k, "measured on another build of this kernel: re-run tools/pmc_refresh.sh and merge the table entries")


def test_symbol_fragment_of_the_run_time_order_kernel():
    assert "ILin1E" in codeobj.contact_kernel_symbol_fragment(-1, True, False, 0, 1)
    assert codeobj.contact_kernel_symbol_fragment(6, True, False, 1, 1) == "pair_contact_kernelILi6ELb1ELb0ELb1ELi1ELb0EE"
    assert codeobj.contact_kernel_symbol_fragment(6, True, False, 1, 1, 1) == "pair_contact_kernelILi6ELb1ELb0ELb1ELi1ELb1EE"

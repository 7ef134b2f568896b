"""What the compiler makes of the places where this round's measurements turned on the generated code (no GPU needed:
hipcc cross-compiles gfx950):
  * the lean build of igd_scan_sorted -- the dominant kernel -- keeps 8 waves per SIMD (<= 64 VGPRs) and spills nothing;
  * its pairwise compare loop is the written-out one (s_bitset0_b64 on VCC, branch on VCC itself: match_slot_asm);
  * the workgroup's LDS counters of the batch's last launch are reached with LDS atomics (ds_add_u64), not with flat atomics
    that resolve to LDS at run time (TailHist);
  * (round 5) the loads of the exact walk, of k_split_fine_a's staged path and of the radix sort's histogram go out back to back,
    not one round trip each (no load behind a divergent branch).
The device code is compiled to assembly once per session (~35 s)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LEAN = "_Z15igd_scan_sortedILb0ELb1ELb1ELb0ELb0ELi0EEv5SortK"


@pytest.fixture(scope="session")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "igd_hip.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S",
                           "--cuda-device-only", os.path.join(ROOT, "igd_amd", "csrc", "igd_hip.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def body(isa, symbol):
    """Instructions of one kernel (from its label to its s_endpgm) and its .amdhsa_ descriptor block."""
    m = re.search(r"^%s\w*:" % re.escape(symbol), isa, re.M)
    assert m, "kernel %s not in the code object" % symbol
    code = isa[m.start():isa.index("s_endpgm", m.start())]
    d = isa.index(".amdhsa_kernel " + symbol)
    return code, isa[d:isa.index(".end_amdhsa_kernel", d)]


def field(desc, name):
    return int(re.search(r"\.amdhsa_%s\s+(\d+)" % name, desc).group(1))


def test_lean_scan_kernel_keeps_eight_waves_and_spills_nothing(isa):
    code, desc = body(isa, LEAN)
    assert field(desc, "private_segment_fixed_size") == 0, "the lean scan kernel uses scratch (spills)"
    assert field(desc, "next_free_vgpr") <= 64, "more than 64 VGPRs: fewer than 8 waves per SIMD"
    assert "scratch_" not in code and "buffer_store_dword" not in code


def test_compare_loop_is_the_written_out_one(isa):
    code, _ = body(isa, LEAN)
    # one loop per slot: s_ff1 on VCC, the bit cleared with s_bitset0, the loop closed by a branch on VCC itself
    assert code.count("s_bitset0_b64 vcc") >= 5
    assert code.count("s_cbranch_vccnz 1b") >= 5
    assert code.count("v_pk_max_u16") >= 10


def test_last_launch_counts_in_lds_with_lds_atomics(isa):
    for sym in ("_Z14k_reduce_slabsILb0EEv5SortK", "_Z14k_reduce_slabsILb1EEv5SortK", "_Z12k_exact_walkILb0EEv5SortK", "_Z12k_exact_walkILb1EEv5SortK"):
        code, _ = body(isa, sym)
        assert "flat_atomic" not in code, sym + ": a flat atomic (an LDS counter reached through a generic pointer?)"
        assert "ds_add_u64" in code, sym + ": no LDS atomic for the workgroup's counters"


def _longest_run_of_loads(code, what):
    """Longest run of vector-memory loads matching `what` with no s_waitcnt vmcnt between them."""
    best = run = 0
    for line in code.split("\n"):
        t = line.strip()
        if re.match(what, t):
            run += 1
            best = max(best, run)
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            run = 0
    return best


def test_no_round_trip_per_load_where_round_5_found_them(isa):
    """LABNOTES R5-12: a load behind a divergent branch (`if (i < n) x = p[i];` with a default in x) is waited for before the next
    one is asked for.  The three places where that cost measurable time must keep their loads back to back:
      * the exact walk of a long query's last tile: five record words + five dataset numbers of a walk in one go (was: a wait per
        slot, five round trips per walk);
      * k_split_fine_a's staged path: a thread's eight tuples (was: eight round trips);
      * the radix sort's histogram: a thread's eight keys."""
    code, _ = body(isa, "_Z14k_reduce_slabsILb0EEv5SortK")
    assert _longest_run_of_loads(code, r"buffer_load_(dword|ushort) ") >= 10, "the walk's loads are waited for one by one again"
    code, _ = body(isa, "_Z14k_split_fine_a")
    assert _longest_run_of_loads(code, r"global_load_dwordx3 ") >= 8, "k_split_fine_a's tuple loads are waited for one by one again"
    code, _ = body(isa, "_ZL10ss_rs_hist")
    assert _longest_run_of_loads(code, r"global_load_dword ") >= 8, "ss_rs_hist's key loads are waited for one by one again"

"""Static guard on the resources of the perf-critical kernels: hipcc cross-compiles to gfx950 ISA without a GPU, and the
kernel descriptors in the assembly say how many VGPRs / how much scratch and LDS each kernel takes.  A change that makes
the tile parser spill, or drops its occupancy below what DESIGN.md §8 records, fails here instead of showing up as a
slower bench line three steps later."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transferia_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


_ASM = {}


def _compile_all():
    """Every source these tests look at, compiled side by side on first use (one hipcc run a file is 10-60 s; one after another they
    were three minutes of the CPU suite)."""
    if _ASM:
        return
    import concurrent.futures
    srcs = sorted(set(re.findall(r'"(tf_[a-z_]+\.hip)"', open(os.path.abspath(__file__)).read())))   # every source a test names

    def one(src):
        return src, subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", "-", os.path.join(CSRC, src)],
                                   capture_output=True, text=True, timeout=900)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        for src, r in ex.map(one, srcs):
            _ASM[src] = r


def kernel_table(src):
    _compile_all()
    asm = _ASM[src]
    assert asm.returncode == 0, asm.stderr[-2000:]
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm.stdout, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))  # noqa: E731
        out[m.group(1)] = {"vgpr": g("next_free_vgpr"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}
    return out


def find(table, fragment):
    hits = [v for k, v in table.items() if fragment in k]
    assert len(hits) == 1, (fragment, list(table))
    return hits[0]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_tile_parser_resources():
    t = kernel_table("tf_csv.hip")
    k = find(t, "csv_parse_regular")
    # 512 threads, 6 waves per SIMD: at most 80 VGPRs, nothing in scratch, three workgroups of LDS per CU (160 KB).  (Round 6: with only SOME of the
    # profiling branches compiled out the allocator spilled five dwords a lane — 20 bytes x 15 M lanes = 240 MB of scratch writes a launch in the
    # WRITE_SIZE counter, no change in time; with all of them out the kernel needs 79 VGPRs and no scratch.  This assertion is what caught it.)
    assert k["scratch"] == 0 and k["vgpr"] <= 80 and k["lds"] * 3 <= 160 * 1024, k
    g = find(t, "csv_parse_tiles_general")  # the rare tiles: same occupancy, a little scratch for the tile loop is fine
    assert g["scratch"] <= 128 and g["vgpr"] <= 80 and g["lds"] * 3 <= 160 * 1024, g
    assert find(t, "csv_count_newlines")["scratch"] == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_serializer_and_sr_resources():
    t = kernel_table("tf_serialize.hip")
    # the sinks live in registers: every emitter is force-inlined (a sink passed by reference to a real call sits in scratch and
    # every put() becomes a load-modify-store there: measured 10.5 ms -> 6.2 ms on configs[3]); what is left is the float
    # formatter's digit buffer on the JSON / CSV paths
    assert find(t, "ser_cell_write")["scratch"] <= 64 and find(t, "ser_cell_len")["scratch"] == 0
    walks = {k: v for k, v in t.items() if "ser_chunk_write" in k or "ser_chunk_len" in k}
    lean = walks.pop(next(k for k in walks if "ser_chunk_len_fast" in k))
    # the lean JSONEachRow length pass (round 6): no emitters, eight waves per SIMD, nothing spilled
    assert lean["scratch"] == 0 and lean["vgpr"] <= 64 and lean["lds"] == 0, lean
    assert find(t, "ser_text_flags")["scratch"] == 0
    assert len(walks) == 6 and min(v["scratch"] for v in walks.values()) == 0 and max(v["scratch"] for v in walks.values()) <= 64
    assert all(v["vgpr"] <= 128 for v in walks.values())  # four waves per SIMD (and four workgroups of 4 x 9 KiB LDS per CU): the walks are latency-bound
    fc = find(t, "ser_fill_const")  # a wave per (row, 2 KiB piece): full occupancy, nothing spilled
    assert fc["vgpr"] <= 32 and fc["scratch"] == 0
    t = kernel_table("tf_srjson.hip")
    # the parse kernel keeps per-depth key spans for the `any` order check in scratch (2 x 128 words) and nothing more
    assert find(t, "sr_parse_frames")["scratch"] <= 2048 and find(t, "sr_count_frames")["scratch"] == 0
    # the cell kernels every cell runs through do not carry the sorting emitter's frame stack (its cells have their own launch)
    for name in ("sr_cell_values", "sr_cell_text"):
        both = [v for k, v in t.items() if name in k]
        assert len(both) == 2
        lean = min(both, key=lambda v: v["scratch"])
        assert lean["scratch"] == 0 and lean["vgpr"] <= 96, (name, lean)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_json_kernels_stay_lean():
    """The kernels every line / cell runs through do not carry the map emitter (any containers, `_rest`): lines and cells that
    need it go through json_parse_listed and the second copy launch, which do."""
    t = kernel_table("tf_json.hip")
    lines, listed = find(t, "json_parse_linesE"), find(t, "json_parse_listedE")
    assert lines["vgpr"] <= 168 and lines["scratch"] <= 320, lines
    assert listed["vgpr"] > lines["vgpr"]
    lean_listed = find(t, "json_parse_listed_lean")  # what the tile parser hands over goes through the lean grammar first
    assert lean_listed["vgpr"] <= 168 and lean_listed["scratch"] <= 320, lean_listed
    # the tile parser: 512 threads at four waves per SIMD (128 VGPRs, a small spill), two workgroups of LDS per CU
    tiles = find(t, "json_parse_tiles")
    assert tiles["vgpr"] <= 128 and tiles["scratch"] <= 128 and tiles["lds"] * 2 <= 160 * 1024, tiles
    sr_tiles = find(kernel_table("tf_srjson.hip"), "sr_parse_tiles")
    assert sr_tiles["vgpr"] <= 128 and sr_tiles["scratch"] <= 128 and sr_tiles["lds"] * 2 <= 160 * 1024, sr_tiles
    copies = {k: v for k, v in t.items() if "json_copy_cells" in k}
    assert len(copies) == 2
    lean = min(copies.values(), key=lambda v: v["vgpr"])
    assert lean["vgpr"] <= 80 and lean["scratch"] == 0, lean


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_debezium_emitter_resources():
    t = kernel_table("tf_dbzemit.hip")
    # the cell kernels carry every converter (40 cases): their sinks stay in registers — the only scratch is the converters' small digit / byte buffers
    # (a 128-bit decimal's 17 bytes, a 40-digit text) — and the fill kernel, which writes most of the bytes, is a handful of registers per lane
    assert find(t, "dbz_cell_len")["scratch"] <= 256 and find(t, "dbz_cell_write")["scratch"] <= 256
    assert find(t, "dbz_cell_len")["vgpr"] <= 160 and find(t, "dbz_cell_write")["vgpr"] <= 160
    assert find(t, "dbz_fill_const")["vgpr"] <= 32 and find(t, "dbz_fill_const")["scratch"] == 0
    assert find(t, "dbz_walk_len")["scratch"] <= 256 and find(t, "dbz_walk_write")["scratch"] <= 256 and find(t, "dbz_walk_write")["vgpr"] <= 160
    assert find(t, "dbz_event_layout")["scratch"] == 0 and find(t, "dbz_event_count")["scratch"] == 0 and find(t, "dbz_event_fill")["scratch"] == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_quick_tile_parsers_fit_three_workgroups_per_cu():
    # 448 threads x 3 workgroups = 21 waves per CU: at most 96 VGPRs, no scratch, a third of the CU's 160 KB of LDS each
    for src, name in (("tf_json.hip", "json_parse_quick"), ("tf_srjson.hip", "sr_parse_quick"), ("tf_debezium.hip", "dbz_parse_quick")):
        k = find(kernel_table(src), name)
        assert k["scratch"] == 0 and k["vgpr"] <= 96 and k["lds"] * 3 <= 160 * 1024, (name, k)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_protobuf_any_kernels_do_not_carry_the_map_emitter():
    """Round 6: map<string, V> and repeated-message support grew the `any` kernels from 124 VGPRs / no scratch to 180 / 184 bytes and cost 15 % of the
    sr_proto line on a schema that has neither; the wide columns have their own instantiation now, and this is the guard."""
    t = kernel_table("tf_protobuf.hip")
    cells = find(t, "pb_cellsILb1ELb0EE")
    assert cells["vgpr"] <= 128 and cells["scratch"] == 0, cells
    text = find(t, "pb_text_anyILb0EE")
    assert text["vgpr"] <= 128 and text["scratch"] <= 64, text
    assert find(t, "pb_cellsILb0ELb0EE")["vgpr"] <= 16

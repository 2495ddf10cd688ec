"""Static checks on the gfx950 instruction streams of the weights-stationary kernels (hipcc cross-compiles without a GPU): the 64 -> 64 kernel of the launch list
(csrc/y7t_conv_ws.hip) and its 128-channel sibling (csrc/y7t_conv_ws128.hip), each in its statically partitioned form and on the tile counter.

The kernel writes its MFMAs as asm statements (accumulators in arch VGPRs, weights in ACC registers), so the compiler's hazard recogniser does not see them:
the first device run of that form copied accumulator registers at the loop exit while the last MFMAs of a tile were still writing them.  What protects the
kernel now is structural -- >= 11 wait states of instructions that touch no accumulator behind every tile's last MFMA -- and this test pins it, together with
the properties the schedule was built for: no packed-fp32 VALU instructions (they hold the matrix pipe off, profiles/r03_issue_classes.txt), no scratch, one
LDS fragment read and at most one transcendental per MFMA slot.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    """VGPR numbers named by an operand token: v7, v[4:7]"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


# file, kernel name, instance pattern, pieces per wave and tile, stores per wave and tile, accumulator registers of one set, MFMAs per wave and tile
KERNELS = {"ws64": ("y7t_conv_ws.hip", "k_conv3x3_c64_ws", r"wsILi\dELi0ELb[01]EE", 13, 8, 64, 144),
           "ws128": ("y7t_conv_ws128.hip", "k_conv3x3_c128_ws", r"wsILi\dELb[01]ELb0EE", 8, 4, 32, 144),
           "ws128_s2": ("y7t_conv_ws128.hip", "k_conv3x3_c128_ws", r"wsILi\dELb[01]ELb1EE", 12, 2, 16, 72)}


@pytest.fixture(scope="module", params=sorted(KERNELS))
def kernels(request, tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    import importlib
    build = importlib.import_module("yolov7_tracker_amd.build")
    fname, kname, inst, npw, nst, nacc, nmf = KERNELS[request.param]
    out = str(tmp_path_factory.mktemp("isa") / (request.param + ".s"))
    src = os.path.join(build.CSRC, fname)
    cmd = [HIPCC] + [f for f in build.FLAGS if f != "-fPIC"] + build.FILE_FLAGS[fname] + ["-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, capture_output=True)
    text = open(out).read()
    ks = {}
    for m in re.finditer(r"^(_ZN\S*%s\w*):[^\n]*\n(.*?)s_endpgm" % kname, text, re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        ks[m.group(1)] = body
    ks = {n: b for n, b in ks.items() if re.search(inst, n)}          # the instances without timing ablations (one per activation, static partition and tile counter)
    assert len(ks) == 6, sorted(ks)
    meta = {n: (int(re.search(r"\.name:\s+%s\n.*?\.private_segment_fixed_size:\s+(\d+)" % re.escape(n), text, re.S).group(1)),
                int(re.search(r"\.name:\s+%s\n.*?\.vgpr_spill_count:\s+(\d+)" % re.escape(n), text, re.S).group(1))) for n in ks}
    return ks, meta, (npw, nst, nacc, nmf)


def test_no_packed_fp32_no_scratch(kernels):
    ks, meta, _ = kernels
    for n, body in ks.items():
        assert not [i for i in body if re.match(r"v_pk_(mul|add|fma)_f32", i)], n
        assert not [i for i in body if i.startswith("scratch_")], n
        # static partition: nothing spilled.  Tile-counter instances: no scratch either; the SiLU one parks two VGPRs (a store base address) in spare ACC registers
        # (v_accvgpr_write / read, eight moves per two tiles of 288 MFMAs)
        assert meta[n][0] == 0 and meta[n][1] <= (2 if _is_dyn(n) else 0), (n, meta[n])


def test_the_tile_counter_fetch_lands_in_registers_the_compiler_does_not_use(kernels):
    """DYN instances (dynamic tile scheduling): the chunk fetch is an asm atomic whose result arrives asynchronously in a255 (a254 carries the increment); it is read back
    behind the counted wait of the next tile body.  Nothing else in the kernel may name those two registers (the compiler knows them only as clobbers of the two asm
    statements), the atomic must not be followed by a wait of the compiler's (`s_waitcnt vmcnt(0)` on the spot is what the builtin atomicAdd produced), and the ring of
    chunk ids is accessed with LDS instructions (a generic pointer made it flat loads with vmcnt(0) waits)."""
    ks, _, _ = kernels
    dyn = {n: b for n, b in ks.items() if _is_dyn(n)}
    assert len(dyn) == 3
    for n, body in dyn.items():
        named = [i for i, ins in enumerate(body) if re.search(r"\ba25[45]\b|a\[\d+:25[45]\]", ins)]
        kinds = [body[i].split()[0] for i in named]
        assert sorted(set(kinds)) == ["global_atomic_add", "v_accvgpr_read_b32", "v_accvgpr_write_b32"], (n, kinds)
        assert kinds.count("global_atomic_add") == 2 and kinds.count("v_accvgpr_read_b32") == 1, (n, kinds)      # first tile + the loop's even body; the loop's odd body reads
        for i in named:
            if body[i].startswith("global_atomic_add"):
                assert "a255" in body[i].split()[1] and "sc0" in body[i]
                nxt = [ins for ins in body[i + 1:i + 12] if ins.startswith("s_waitcnt")]
                assert not [w for w in nxt if "vmcnt(0)" in w], (n, nxt)
        assert not [ins for ins in body if ins.startswith("flat_")], n
    for n, body in ks.items():
        if n not in dyn:
            assert not [ins for ins in body if re.search(r"\ba25[45]\b", ins) or ins.startswith("global_atomic")], n


def _is_dyn(name):
    """the tile-counter instances: k_conv3x3_c64_ws<ACT, 0, true>, k_conv3x3_c128_ws<ACT, true, S2>"""
    return bool(re.search(r"c64_wsILi\dELi0ELb1EE|c128_wsILi\dELb1ELb[01]EE", name))


def _bodies(body):
    bars = [i for i, ins in enumerate(body) if ins.startswith("s_barrier")]
    assert len(bars) == 3, len(bars)                                              # first tile, and the two alternating bodies of the loop
    for bi, b0 in enumerate(bars):
        yield bi, b0, body[b0:(bars[bi + 1] if bi + 1 < len(bars) else len(body))]


def test_tile_bodies_have_the_spelled_out_slot_structure(kernels):
    ks, _, (npw, nst, _, nmf) = kernels
    for n, body in ks.items():
        for bi, _, seg in _bodies(body):
            mf = [i for i, ins in enumerate(seg) if ins.startswith("v_mfma_f32_32x32x16_f16")]
            assert len(mf) == nmf, (n, bi, len(mf))
            seg = seg[:mf[-1] + 1]
            assert sum(ins.startswith("ds_read_b128") for ins in seg) == nmf                      # a few up front + one behind each MFMA until the tile's last substeps
            assert sum(ins.startswith("buffer_load_dwordx4") for ins in seg) == npw               # the pieces of tile t + 2
            assert sum(ins.startswith("global_store_dwordx4") for ins in seg) == (0 if bi == 0 else nst)
            # a slot holds one transcendental and at most two other VALU instructions; the MFMA may sit anywhere inside its slot, so between two
            # consecutive MFMAs of the steady state (slots 8 .. 135) there are at most two slots' worth
            for a, b in zip(mf[8:nmf - 8], mf[9:nmf - 7]):
                slot = seg[a + 1:b]
                assert sum(ins.startswith(("v_exp_f32", "v_rcp_f32")) for ins in slot) <= 2, (n, bi, slot)
                assert sum(ins.startswith("v_") and not ins.startswith(("v_exp_f32", "v_rcp_f32")) for ins in slot) <= 6, (n, bi, slot)


def test_nothing_touches_an_accumulator_for_eleven_wait_states_behind_a_tiles_last_mfma(kernels):
    ks, _, (_, _, nacc, _) = kernels
    for n, body in ks.items():
        for bi, b0, seg in _bodies(body):
            mf = [i for i, ins in enumerate(seg) if ins.startswith("v_mfma_f32_32x32x16_f16")]
            acc = set()
            for i in mf[-4:]:
                acc |= _regs(seg[i].split()[1].rstrip(","))                        # the accumulators of the body (vdst of its last four MFMAs)
            assert len(acc) == nacc
            waited = 0
            for ins in body[b0 + mf[-1] + 1:]:                                      # follow the fall-through path past the body's end
                if waited >= 11:
                    break
                ops = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", ins)
                touched = set().union(*[_regs(o) for o in ops]) if ops else set()
                assert not (touched & acc), (n, bi, ins)
                m = re.match(r"s_nop (\d+)", ins)
                waited += (int(m.group(1)) + 1) if m else 1
            assert waited >= 11

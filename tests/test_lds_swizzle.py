"""The chunk swizzle of the attention kernels' 128-byte-row LDS images (csrc/attn_fused.hip `key_d`) against the bank model of
tools/hwprobe/lds_sim.py (lane groups and bank moduli per instruction from MI355X_MICROARCH.md): the three ways those images are read -
16 consecutive rows, the key-dealt rows of the transposed kernels, transposed ds_read_b64_tr_b16 fragments - are conflict-free with the key in
the source, and two of them were not with the key it replaced (the counters that confirmed it on hardware: profiles/r05_lds_conflicts_*)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim():
    spec = importlib.util.spec_from_file_location("lds_sim", os.path.join(ROOT, "tools", "hwprobe", "lds_sim.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _source_key():
    src = open(os.path.join(ROOT, "tensorflowasr_amd", "csrc", "attn_fused.hip")).read()
    m = re.search(r"int key_d\(int row\) \{ return ([^;]+); \}", src)
    assert m, "key_d not found"
    expr = m.group(1)
    assert expr == "(row & 3) | (((row >> 3) & 1) << 2)", expr  # (the model below evaluates exactly this expression)
    return lambda row: (row & 3) | (((row >> 3) & 1) << 2)


def test_attention_swizzle_is_conflict_free_in_the_bank_model():
    sim, key = _sim(), _source_key()
    dealt = lambda jt: (lambda r: 32 * (jt >> 1) + (r >> 2) * 8 + (jt & 1) * 4 + (r & 3))
    for kk in range(2):
        for t in range(8):  # consecutive rows: ds_read_b128, four lane groups, one LDS cycle each
            assert sim.frag_rows(lambda r: t * 16 + r, lambda g: kk * 4 + g, key) == 4
        for jt in range(4):  # the rows of a 32-key group dealt to two MFMA tiles
            assert sim.frag_rows(dealt(jt), lambda g: kk * 4 + g, key) == 4
    for n in range(4):
        for base in (0, 32, 64, 96):  # transposed fragments: two ds_read_b64_tr_b16, two lane groups each
            assert sim.frag_kt(n * 16, lambda g: base + g * 8, key) == 4


def test_the_key_it_replaced_conflicted_two_ways():
    sim = _sim()
    old = sim.key_d  # (row >> 1) & 7
    dealt = lambda r: (r >> 2) * 8 + (r & 3)
    assert sim.frag_rows(lambda r: r, lambda g: g, old) == 4
    assert sim.frag_rows(dealt, lambda g: g, old) == 8
    assert sim.frag_kt(0, lambda g: g * 8, old) == 8

"""CPU test of the MXFP8 fragment order (csrc/mx_quant.h, called through the C ABI's dtk_mx_layout — host arithmetic, no GPU):
it must be the INVERSE of the operand map v_mfma_scale_f32_16x16x128_f8f6f4 was measured to have (tools/probe/mx_probe.hip,
profiles/r04_mx_probe.txt) — lane l = (g = l >> 4, i = l & 15), operand byte p: k = 64 (p >> 4) + 16 g + (p & 15) of the instruction's
128; scale block b = k >> 5 taken from lane 16 b + i.  Round 4's first version used ck_tile's descriptor (32 consecutive k per lane)
instead: bit-exact quantiser tests, every GEMV 30 % off.  Also pins the numpy un-packer of the GPU tests to the C code."""
import ctypes as C

import numpy as np
import pytest

from detikzify_amd import _lib


def hw_k(lane, p):
    return 64 * (p >> 4) + 16 * (lane >> 4) + (p & 15)


def hw_scale_lane(lane, p):
    return 16 * (hw_k(lane, p) >> 5) + (lane & 15)


@pytest.fixture(scope="module")
def layout():
    lib = _lib.load_library()

    def f(G, slot, k):
        d, s = C.c_int64(), C.c_int64()
        assert lib.dtk_mx_layout(G, slot, k, C.byref(d), C.byref(s)) == 0
        return d.value, s.value
    return f


def _split(data_off):
    piece, within = divmod(data_off, 1024)
    lane, byte = divmod(within, 16)
    return piece >> 1, piece & 1, lane, byte            # (step * 4 + tile, half, lane, byte)


def test_groups_of_32_are_the_instructions_scale_blocks(layout):
    K = 1024
    seen = set()
    for slot in (0, 5, 17, 33, 63):
        for k in range(K):
            d, s = layout(32, slot, k)
            seen.add(d)
            st, half, lane, byte = _split(d)
            assert st == (k >> 7) * 4 + (slot >> 4) and (lane & 15) == (slot & 15)
            p = half * 16 + byte                        # the kernel builds a lane's operand from its two 16-byte pieces: bytes 0..15 | 16..31
            assert hw_k(lane, p) == (k & 127), (slot, k)
            row, within = divmod(s, 1024)
            tile, rest = divmod(within, 256)
            slane, sbyte = divmod(rest, 4)
            assert row == (k >> 9) and tile == (slot >> 4) and sbyte == ((k >> 7) & 3)
            assert slane == hw_scale_lane(lane, p), "the group's E8M0 byte sits in the lane the instruction reads block k >> 5 from"
    assert len(seen) == 5 * K


def test_groups_of_16_ride_in_every_second_lane_group(layout):
    """down's input: a pair step (128 k) = two instructions s = 0, 1; in instruction s only the lane groups g with (g & 1) == s carry
    weights (the kernel zeroes the others), so each of its four scale blocks holds ONE 16-group: its 16 values sit in one lane's
    16-byte half, and its E8M0 byte in the lane the instruction reads that block's scale from, byte (ps & 1) * 2 + s"""
    K = 1024
    for slot in (0, 9, 30, 63):
        for grp in range(K // 16):
            where = set()
            for k in range(grp * 16, grp * 16 + 16):
                d, s = layout(16, slot, k)
                st, half, lane, byte = _split(d)
                ps, sub, q = k >> 7, (k >> 6) & 1, (k >> 4) & 3
                assert st == ps * 4 + (slot >> 4) and (lane & 15) == (slot & 15) and byte == (k & 15)
                assert ((lane >> 4) & 1) == sub, "a value of instruction s sits in a lane group the kernel leaves alive for s"
                p = half * 16 + byte
                assert hw_k(lane, p) >> 5 == q, "its scale block in that instruction is its 16-group's index"
                row, within = divmod(s, 1024)
                tile, rest = divmod(within, 256)
                slane, sbyte = divmod(rest, 4)
                assert row == (ps >> 1) and tile == (slot >> 4) and sbyte == (ps & 1) * 2 + sub
                assert slane == hw_scale_lane(lane, p)
                where.add((st, half, lane, s))
            assert len(where) == 1, "a 16-group = one lane's 16-byte half and one scale byte"
    # the two instructions of a pair step together fill every lane of the 2 KiB tile exactly once
    offs = {layout(16, 3, k)[0] for k in range(128)}
    assert len(offs) == 128 and {o // 1024 for o in offs} == {0, 1}


def test_the_gpu_tests_unpacker_is_the_c_layout(layout):
    from tests.test_gpu_parity_mx import mx_unpack
    for G, K in ((32, 1024), (16, 768)):
        nbytes = -(-K // 128) * 4 * 2048
        x8 = np.arange(nbytes, dtype=np.int64)          # every byte holds its own offset
        xs = np.arange(-(-K // (16 * G)) * 1024, dtype=np.int64)
        codes, scales = mx_unpack(x8, xs, K, G, 64)
        for slot in (0, 21, 47, 63):
            for k in range(0, K, 7):
                d, s = layout(G, slot, k)
                assert codes[slot, k] == d and scales[slot, k // G] == s

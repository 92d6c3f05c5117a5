"""CPU: the integer arithmetic by which the persistent NT GEMM shares out its remainder round by K range (csrc/gemm.hip: nt_stream_k on
the host, the item set-up at the top of gemm_bf16_nt256_kernel<MODE, SK = true> on the device), restated in Python and checked for
every tile count, K and grid the host rule admits: each remainder tile is covered exactly once by contiguous pieces of at least two
stage pairs, its first piece belongs to the owner, the owner's follower count matches, the followers are the next workgroups of
the owner's XCD (blockIdx + 8, + 16, ...), no workgroup's range reaches into a third tile, and every packed field fits its bits.
The GPU parity tests (tests/test_kernels_gpu.py::test_gemm_nt_stream_k_remainder) run a handful of these configurations; this
sweep covers the rest of the space the dispatch can reach (other models' shapes, reserved-CU grids)."""
import pytest


def host_rule(tiles, K, G, gain=20, min_k=512):
    P = K // 64
    if G > 256 or K < min_k or P < 8 or P > 1023:
        return None
    rem = tiles % G
    if rem == 0 or (G - rem) * P < gain * G:
        return None
    gx, nxmax = G // 8, (rem + 7) // 8
    if G % 8 or nxmax * P > gx * (P - 1):
        return None
    return rem, min(gx, max(1, (rem // 8) * P // 4)), P


def device_items(G, sk_tiles, sk_wgs, P):
    items = {}
    for blk in range(G):
        x, j = blk & 7, blk >> 3
        t0 = (x * sk_tiles) >> 3
        nx = (((x + 1) * sk_tiles) >> 3) - t0
        wx = min(sk_wgs, nx * 4)
        first = owner = None
        if j < wx:
            def bound(c):
                if c >= wx:
                    return nx * P
                v = (c * nx * P) // wx
                r = v % P
                return v - 1 if r == 1 else (v + 1 if r == P - 1 else v)
            b, en = bound(j), bound(j + 1)
            t = b // P
            off = b - t * P
            if off:
                pe = min(en, (t + 1) * P)
                first = (t0 + t, off, pe - b, 0)
                b, t = pe, t + 1
            if b < en:
                tend = (t + 1) * P
                assert en <= tend, "a range reaches into a third tile"
                nf, c2 = 0, j + 1
                while c2 < wx and bound(c2) < tend:
                    nf, c2 = nf + 1, c2 + 1
                owner = (t0 + t, 0, min(en, tend) - b, nf)
        items[blk] = (first, owner)
    return items


def check(G, sk_tiles, sk_wgs, P):
    cover = {}
    for blk, pieces in device_items(G, sk_tiles, sk_wgs, P).items():
        for it in pieces:
            if it is None:
                continue
            t, off, n, nf = it
            assert 0 <= t < sk_tiles and t < 256 and off < 1024 and 2 <= n <= 1023 and nf <= 7 and off + n <= P
            cover.setdefault(t, []).append((off, n, blk, nf))
    assert sorted(cover) == list(range(sk_tiles))
    for t, pcs in cover.items():
        pcs.sort()
        pos = 0
        for off, n, _, _ in pcs:
            assert off == pos
            pos += n
        assert pos == P
        assert pcs[0][3] == len(pcs) - 1  # the owner (first K range) collects exactly the other pieces ...
        assert [p[2] for p in pcs[1:]] == [pcs[0][2] + 8 * (i + 1) for i in range(len(pcs) - 1)]  # ... from the next workgroups of its XCD


@pytest.mark.parametrize("G", [256, 240, 192, 64])
def test_stream_k_partition_covers_every_remainder_tile_once(G):
    checked = 0
    for K in (512, 768, 1024, 2304, 3072, 4096, 65472):
        for tiles in list(range(1, 2 * G + 2, 3)) + [651, 2604, 1953, 165, 495, 660, 132, 396]:
            for gain in (1, 20):
                h = host_rule(tiles, K, G, gain)
                if h is not None:
                    check(G, *h)
                    checked += 1
    assert checked > 100

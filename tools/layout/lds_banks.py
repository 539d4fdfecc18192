""" LDS bank-conflict check of the split-bf16 activation planes (MI355X_MICROARCH.md, LDS table): row stride RS bytes per
(stream, point) row of 64 bf16 units; accesses: ds_write_b64 of a lane's 4 units, ds_read_b128 of 8 units along K,
ds_read_b64_tr_b16 of 4 points x 16 units (weight-gradient operands). Prints the worst multiplicity per lane group. """
import sys

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def worst(groups, addr, width, nbanks):
    """ addr(lane) -> byte address; width bytes per lane; -> max number of distinct addresses on one bank within a group """
    w = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr(l)
            for d in range(width // 4):
                banks.setdefault(((a // 4) + d) % nbanks, set()).add((a // 4) + d)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def check(rs, T=16):
    out = {}
    # write: lane (lr, lq) of wave w: row lr, byte 32 w + 8 lq; contiguous 16-lane groups, 32 banks
    out['write_b64'] = max(worst([list(range(16 * g, 16 * g + 16)) for g in range(4)],
                                 lambda l, w=w: (l & 15) * rs + 32 * w + 8 * (l >> 4), 8, 32) for w in range(4))
    # read b128 along K: row lr, byte 64 kb + 16 lq
    out['read_b128'] = max(worst(B128_GROUPS, lambda l, kb=kb: (l & 15) * rs + 64 * kb + 16 * (l >> 4), 16, 64) for kb in range(2))
    # transpose read: lane i of group lq reads row pt0(lq) + i // 4, bytes 32 o + 8 (i % 4); 2 x 32 lanes, 64 banks
    def tr(pt_of):
        return max(worst([list(range(0, 32)), list(range(32, 64))],
                         lambda l, o=o: (pt_of(l >> 4) + (l & 15) // 4) * rs + 32 * o + 8 * (l & 3), 8, 64) for o in range(4))
    if T == 16:
        # k-slot (lq, e): stream 2 kb + (lq >> 1) -> another row block (S*T rows apart: offset T * rs, same for the pair), points 4 (lq & 1) + e, 8 + ...
        out['tr_first'] = tr(lambda lq: (lq >> 1) * T + 4 * (lq & 1))
        out['tr_second'] = tr(lambda lq: (lq >> 1) * T + 8 + 4 * (lq & 1))
    else:
        out['tr_first'] = tr(lambda lq: 4 * lq)
        out['tr_second'] = tr(lambda lq: 16 + 4 * lq)
    return out


if __name__ == '__main__':
    for T in (16, 32):
        for rs in (128, 136, 144, 152, 160, 176, 192, 208):
            print(T, rs, check(rs, T))


def check_swz(rs, f, T=16):
    """ same accesses with the 16-byte chunk index of a row XORed by f(row) """
    def A(row, byte):
        return row * rs + ((((byte >> 4) ^ f(row)) & 7) << 4) + (byte & 15)
    out = {}
    out['write_b64'] = max(worst([list(range(16 * g, 16 * g + 16)) for g in range(4)],
                                 lambda l, w=w: A(l & 15, 32 * w + 8 * (l >> 4)), 8, 32) for w in range(4))
    out['read_b128'] = max(worst(B128_GROUPS, lambda l, kb=kb: A(l & 15, 64 * kb + 16 * (l >> 4)), 16, 64) for kb in range(2))
    def tr(pt_of):
        return max(worst([list(range(0, 32)), list(range(32, 64))],
                         lambda l, o=o: A(pt_of(l >> 4) + (l & 15) // 4, 32 * o + 8 * (l & 3)), 8, 64) for o in range(4))
    if T == 16:
        out['tr_first'] = tr(lambda lq: (lq >> 1) * T + 4 * (lq & 1))
        out['tr_second'] = tr(lambda lq: (lq >> 1) * T + 8 + 4 * (lq & 1))
    else:
        out['tr_first'] = tr(lambda lq: 4 * lq)
        out['tr_second'] = tr(lambda lq: 16 + 4 * lq)
    return out


def search():
    fs = {'0': lambda r: 0, 'r&7': lambda r: r & 7, '(r>>1)&7': lambda r: (r >> 1) & 7, '(r>>2)&3': lambda r: (r >> 2) & 3,
          '(r>>1)&3': lambda r: (r >> 1) & 3, 'r&3': lambda r: r & 3, '(r&7)^(r>>3)': lambda r: (r & 7) ^ ((r >> 3) & 1),
          '2*(r&3)': lambda r: 2 * (r & 3), '(r>>3)&1': lambda r: (r >> 3) & 1, '((r>>3)&1)*2': lambda r: ((r >> 3) & 1) * 2,
          '((r>>2)&1)*4': lambda r: ((r >> 2) & 1) * 4, '(r>>2)&1': lambda r: (r >> 2) & 1}
    for T in (16, 32):
        for rs in (128, 144, 160, 176):
            for name, f in fs.items():
                c = check_swz(rs, f, T)
                if max(c.values()) <= 2 and sum(c.values()) <= 5:
                    print('T', T, 'RS', rs, 'f =', name, c)


if __name__ == '__main__':
    print('--- swizzle search')
    search()

import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint8)
n_steps, n_roles, grid, ns = [int(x) for x in raw[:16].view(np.int32)]
kinds = raw[16:16 + 4 * n_roles].view(np.int32)
st = raw[16 + 4 * n_roles:].view(np.uint64).reshape(n_steps, n_roles, ns).astype(np.float64)
st[st == 0] = np.nan
kind, layer, row = kinds & 0xff, (kinds >> 8) & 0xff, kinds >> 16
sel = st[8:]
for k, nm, order in ((0, "attn", (1, 2, 4, 3, 5, 6)), (1, "cross", (1, 2, 4, 5, 3, 6))):
    for l in sorted(set(layer[kind == k])):
        cols = (kind == k) & (layer == l) & (row == 0)
        s = sel[:, cols, :]
        out = []
        for a, b in zip(order[:-1], order[1:]):
            out.append("%d->%d %.2f" % (a, b, np.nanmean((s[:, :, b] - s[:, :, a]) * 0.01)))
        print(nm, "L%d" % l, "  ".join(out))

""" WHERE do two runs of the same fused step differ? Runs the step `--reps` times on one build of the library, keeps the per-workgroup
partial gradient rows (the head of the workspace: [grid][p_total]) of every run and reports, against the first run, which workgroups and
which parameter blocks (W1, b1, hidden W / b per layer, WL, bL, log_scale, loss) hold different bits.
Usage: python tools/diff_runs.py cfg2 lib.so [--gemm bf16x3] [--cap 0] [--reps 6] """
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
import pinn_configs as pc   # noqa: E402
import pydens_amd as pa     # noqa: E402
from pydens_amd import engine   # noqa: E402


def blocks(lay):
    out = [('W1', lay.off_w1, lay.off_b1), ('b1', lay.off_b1, lay.off_wh)]
    for l in range(lay.lh):
        o = lay.off_wh + l * lay.hidden_stride
        out += [(f'W{l + 2}', o, o + lay.hp * lay.hp), (f'b{l + 2}', o + lay.hp * lay.hp, o + lay.hidden_stride)]
    out += [('WL', lay.off_wl, lay.off_bl), ('bL', lay.off_bl, lay.off_bl + 1), ('log_scale', lay.off_log_scale, lay.off_log_scale + 1),
            ('loss', lay.off_loss, lay.off_loss + 1)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workload')
    ap.add_argument('lib')
    ap.add_argument('--gemm', default='bf16x3')
    ap.add_argument('--cap', type=int, default=0)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--net', action='store_true', help='-DPINN_DUMP_NET builds: compare the network outputs point by point')
    ap.add_argument('--mt', type=int, default=1, help='16-point row tiles per tile of the kernel (for the report)')
    ap.add_argument('--brief', action='store_true')
    ap.add_argument('--save', default=None, help='prefix of .npz files with the per-lane dumps of the differing (tile, wave) pairs')
    args = ap.parse_args()
    lib = engine.bind(ctypes.CDLL(args.lib))
    torch.manual_seed(0)
    cfg = pc.make_config(args.workload, pa.D, torch, V=pa.V)
    solver = pa.Solver(cfg['equation'], **cfg['solver_kwargs'], _lib=lib)
    lib.pinn_debug_max_wgs_per_cu(solver.model.net.handle, args.cap)
    solver.set_gemm_mode(args.gemm)
    n = min(cfg['n_points'], 131072)
    xs = torch.from_numpy(pc.sample_points(cfg, n, seed=1)).cuda()
    lay = solver.model.net.layout
    spec = solver.spec
    rows, nets, layers, lanes = [], [], [], []
    dump = None
    if args.net:
        dump = torch.zeros(8 * n + 64 * (n // 16 + 1) + 2048 * (n // 16 + 1), dtype=torch.float32, device='cuda')
        lib.pinn_debug_phase_buffer(ctypes.c_void_p(dump.data_ptr()))
    for r in range(args.reps):
        solver.grads.zero_()
        solver._fused_step(xs, 1)
        torch.cuda.synchronize()
        info = (ctypes.c_int32 * 4)()
        lib.pinn_last_launch_info(info)
        ws = solver.model.workspace(n, spec.nd, spec.n2p)
        rows.append(ws[:info[0] * lay.p_total].view(info[0], lay.p_total).cpu().numpy().copy())
        if dump is not None:
            nets.append(dump[:8 * n].view(8, n).cpu().numpy().copy())
            layers.append(dump[8 * n:8 * n + 64 * (n // (16 * args.mt))].view(-1, 16, 4).cpu().numpy().copy())
            o = 8 * n + 64 * (n // (16 * args.mt) + 1)
            lanes.append(dump[o:o + 2048 * (n // (16 * args.mt))].view(-1, 4, 2, 64, 4).cpu().numpy().copy())
    print(f'{os.path.basename(args.lib)} {args.workload} {args.gemm}: {lib.pinn_last_kernel_name().decode()} grid {info[0]} = {info[1]}/CU x {info[2]} threads')
    if nets:
        # which points' network outputs differ, and where do they sit? tile = 16 * mt points; workgroup = tile % grid (one team)
        T = 16 * args.mt
        for r in range(1, args.reps):
            d = nets[r].view(np.uint32) != nets[0].view(np.uint32)
            pts_ = np.nonzero(d.any(axis=0))[0]
            print(f'run {r}: network output differs at {len(pts_)} of {n} points; streams touched: {np.nonzero(d.any(axis=1))[0].tolist()}')
            for p_ in pts_[:4]:
                tile = p_ // T
                rel = [abs(float(nets[r][s_, p_]) - float(nets[0][s_, p_])) / max(abs(float(nets[0][s_, p_])), 1e-30) for s_ in range(8) if d[s_, p_]]
                print(f'    point {p_}: tile {tile} (workgroup {tile % info[0]}, its tile #{tile // info[0]}), row {p_ % T} of the tile; streams '
                      f'{np.nonzero(d[:, p_])[0].tolist()} rel. change {", ".join(f"{x:.1e}" for x in rel)}')
            # did a differing tile work on the points of another tile? (dump rows S, S + 1 = the coordinates the point stage saw)
            S_ = spec.nd + 2 if solver.residual_plan is not None and solver.residual_plan.comb_w is not None else spec.n_streams
            xs_host = xs.cpu().numpy()
            for tile in sorted(set((pts_ // T).tolist()))[:40]:
                sl = slice(tile * T, tile * T + T)
                for which, data in (('run 0', nets[0]), (f'run {r}', nets[r])):
                    seen = data[S_:S_ + 2, sl].T
                    if not np.array_equal(seen, xs_host[sl, :2]):
                        src = [t2 for t2 in range(n // T) if np.array_equal(seen, xs_host[t2 * T:t2 * T + T, :2])]
                        print(f'    tile {tile} (workgroup {tile % info[0]}, its tile #{tile // info[0]}) in {which}: worked on the points of tile '
                              f'{src if src else "?? (no tile of the batch)"}' + (f' = its own tile #{src[0] // info[0]}' if src and src[0] % info[0] == tile % info[0] else ''))
            # per differing tile: which of the dumped per-wave checksums differ (slots 0..3: value stream behind dense layer 1..4; 4 point
            # row, 5 bias, 6 weight rows, 7 saved value of layer 1, 8.. derivative streams of layer 1)
            ld = layers[r].view(np.uint32) != layers[0].view(np.uint32)
            names = ['h1', 'h2', 'h3', 'h4', 'x', 'b1', 'W1', 'sv1', 'h1_s1', 'h1_s2', 'h1_s3', 'h1_s4']
            firsts = {}
            for tile in np.nonzero(ld.any(axis=(1, 2)))[0]:
                key = tuple((names[k] if k < len(names) else str(k), tuple(np.nonzero(ld[tile, k])[0].tolist())) for k in range(16) if ld[tile, k].any())
                firsts.setdefault(key, []).append(int(tile))
            for key, tiles in sorted(firsts.items(), key=lambda kv: -len(kv[1]))[:12]:
                print(f'    {len(tiles):3d} tiles (e.g. {tiles[:4]}): ' + ', '.join(f'{nm} waves {list(w)}' for nm, w in key))
            # the first layer per lane: which lanes (lr = lane & 15 is the point row, lq = lane >> 4 the unit quad) and which of the
            # lane's four units hold different values
            dl = lanes[r].view(np.uint32) != lanes[0].view(np.uint32)
            if args.save:
                idx = np.nonzero(dl.any(axis=(2, 3, 4)))
                np.savez(f'{args.save}_run{r}.npz', tiles=idx[0], waves=idx[1], a=lanes[0][idx], b=lanes[r][idx],
                         xs=np.stack([xs_host[t * T:(t + 1) * T] for t in idx[0]]) if len(idx[0]) else np.zeros((0, T, 2)))
            shown = 0
            for tile, wave_ in zip(*np.nonzero(dl.any(axis=(2, 3, 4)))):
                if shown >= 10:
                    break
                shown += 1
                for st in range(2):
                    d_ = dl[tile, wave_, st]
                    if d_.any():
                        ln = np.nonzero(d_.any(axis=1))[0]
                        a_, b_ = lanes[0][tile, wave_, st][d_], lanes[r][tile, wave_, st][d_]
                        print(f'    tile {tile} wave {wave_} stream {st}: {int(d_.sum())} of 256 values differ; lanes {ln.tolist()[:40]}{"..." if len(ln) > 40 else ""}; '
                              f'unit index within the lane {sorted(set(np.nonzero(d_)[1].tolist()))}; e.g. {a_[:3].tolist()} vs {b_[:3].tolist()}')
            if len(pts_):
                print(f'    rows within the tile: {sorted(set((pts_ % T).tolist()))}; tile numbers within the workgroup: {sorted(set(((pts_ // T) // info[0]).tolist()))}')
    ref = rows[0]
    for r in range(1, args.reps):
        diff = rows[r].view(np.uint32) != ref.view(np.uint32)
        wgs = np.nonzero(diff.any(axis=1))[0]
        print(f'run {r}: {int(diff.sum())} of {diff.size} partial-row entries differ from run 0, in {len(wgs)} of {diff.shape[0]} workgroups'
              + (f' (first: {wgs[:12].tolist()})' if len(wgs) else ''))
        for name, a, b in blocks(lay):
            if args.brief:
                break
            d = diff[:, a:b]
            if d.any():
                w = np.nonzero(d.any(axis=1))[0]
                cols = np.nonzero(d.any(axis=0))[0]
                rel = np.abs(rows[r][:, a:b][d].astype(np.float64) - ref[:, a:b][d]) / np.maximum(np.abs(ref[:, a:b][d]), 1e-30)
                extra = ''
                if name.startswith('W') and name not in ('W1', 'WL'):
                    out_rows = sorted(set((cols // lay.hp).tolist()))
                    in_cols = sorted(set((cols % lay.hp).tolist()))
                    extra = f'; output units {out_rows[:20]}{"..." if len(out_rows) > 20 else ""}, input units {in_cols[:20]}{"..." if len(in_cols) > 20 else ""}'
                print(f'    {name:9s} {int(d.sum()):6d} entries in {len(w):3d} workgroups; relative change median {np.median(rel):.1e} max {rel.max():.1e}{extra}')


if __name__ == '__main__':
    main()

"""hunt the end-of-launch stall of the pipeline (error bit 128): long launches of Azul at 1600 simulations until one times out, then the post-mortem:
what the descent wave that gave up saw (library built with -DAZG_ASYNC_POSTMORTEM) against what is in memory after the launch.
usage: AZG_LIB=build_ab/libazg_pm.so [AZG_ASYNC_TIMEOUT_MS=3000] python tools/dbg_async_stall.py [launches]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from azg_amd import games, nnet, _lib
from azg_amd.selfplay import SelfPlayEngine
class Args(dict): __getattr__ = dict.get
T, SIMS, ROUNDS, CAP = 4096, int(os.environ.get('SIMS', 1600)), int(os.environ.get('ROUNDS', 96000)), int(os.environ.get('CAP', 44000))
G = os.path.join(ROOT, 'tests/golden')
a = Args(numMCTSSims=SIMS, cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True, dirichletAlpha=-1, temperature=[1.25, 0.8, 1.0], tempThreshold=10, ratio_fullMCTS=5, prob_fullMCTS=1.0)
g = games.AzulGame()
net = nnet.MobileNet1dHip(nnet.AzulV84.from_npz(G + '/weights_azul_v84.npz', device='cuda:0'), max_batch=T)
e = SelfPlayEngine(g, net, a, T, node_capacity=CAP, max_examples=T * 160, use_graph=False)
e.start(); e.run(5 * SIMS); torch.cuda.synchronize()
f = e.forest
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    f.async_counters(reset=True)
    t0 = time.perf_counter(); e.run(ROUNDS); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = f.async_profile(reset=False)
    print('launch %d: %.2f s, timeouts %s, abort %d retired %d, engine errors %d, plies %d' % (rep, dt, p['timeouts'], p['ctl']['abort'], p['ctl']['retired'], e.stats()['errors'], e.stats()['plies']), flush=True)
    if (p['timeouts']['select'] or p['timeouts']['net'] or p['ctl']['abort']) and os.environ.get('POSTMORTEM'):
        n_sel, n_net = p['n_sel'], p['n_net']
        out = (C.c_uint64 * (792 + 104 * 40))(); ready = (C.c_uint32 * (128 * 256))(); ring = (C.c_uint64 * (1 << 16))()
        rb = _lib.check(_lib.lib().azg_forest_async_debug(f.h, out, ready, 256, ring, 1 << 16))
        mask = (1 << rb) - 1
        tail, head = p['ctl']['leaf_tail'], p['ctl']['leaf_head']
        ne = f.needs_eval.cpu().numpy()
        print('  abort code %d; trees whose last call queued a leaf (needs_eval): %d' % (p['ctl']['abort'], int(ne.sum())))
        # every ring slot: (ticket of the current lap that maps there, entry)
        by_tree = {}
        for s_ in range(1 << rb):
            ent = int(ring[s_])
            if ent == 0: continue
            t = ent & 0xFFFFF; left = (ent >> 32) & 0xFFFFFF; tag = ent >> 60
            # the most recent ticket <= tail - 1 that maps to this slot and carries this tag
            lap_now = (tail - 1) >> rb
            cand = None
            for lap in (lap_now, lap_now - 1, lap_now - 2):
                if lap >= 0 and ((lap & 7) + 1) == tag:
                    cand = (lap << rb) | s_; break
            by_tree.setdefault(t, []).append((cand, left, tag, s_))
        held = {}
        for b in range(n_net):
            base, taken, valid, lastn = int(out[280 + 2 * b] & 0xFFFFFFFF), int(out[280 + 2 * b] >> 32), int(out[281 + 2 * b] & 0xFFFFFFFF), int(out[281 + 2 * b] >> 32)
            held[base] = (b, taken, valid, lastn)
        wi = f.async_wginfo(reset=False)
        n_lost = 0
        for t in range(T):
            if not ne[t]: continue
            gi, ii = t % n_sel, t // n_sel
            word = int(ready[gi * 128 + ii])
            ents = sorted(by_tree.get(t, []), key=lambda x: -(x[0] or -1))
            newest = ents[0] if ents else None
            state = 'no ring entry left' if newest is None else ('handed back (ready word %d == left + 1)' % word if word == newest[1] + 1 else
                     'NOT handed back: ring entry ticket %s left %d (tag %d, slot %d), ready word in memory %d' % (newest[0], newest[1], newest[2], newest[3], word))
            if newest is None or word != newest[1] + 1:
                n_lost += 1
                if n_lost <= 16:
                    tk = newest[0] if newest else None
                    hb = held.get(tk & ~15) if tk is not None else None
                    print('  tree %d (workgroup %d, index %d): %s%s' % (t, gi, ii, state, '' if tk is None else '; ticket %s tail (tail %d, head %d), range base %d held at exit by net workgroup %s' % ('<' if tk < tail else '>=', tail, head, tk & ~15, None if hb is None else '%d (taken %04x valid %d last batch %d; batches %d, stayed %.0f us)' % (hb[0], hb[1], hb[2], hb[3] if hb[3] < 1 << 31 else hb[3] - (1 << 32), wi[n_sel + hb[0]][5], wi[n_sel + hb[0]][7]))))
        print('  queued leaves never handed back: %d' % n_lost)
        for b in range(min(n_net, 104)):
            dd = out[792 + 40 * b: 792 + 40 * (b + 1)]
            if dd[4]:
                print('  net workgroup %d was GIVEN an implausible range: base %d while the tail was %d (previous base %d) at %.4f s; head re-read %d; batches so far %d' % (
                    b, dd[4] & 0xFFFFFFFF, dd[4] >> 32, dd[5] & 0xFFFFFFFF, (dd[5] >> 32) / 1e8, dd[6] & 0xFFFFFFFF, dd[6] >> 32))
            if dd[0] == 0 and dd[1] == 0: continue
            if (dd[2] / 1e8) < 0.01: continue
            base, taken = int(dd[0] & 0xFFFFFFFF), int(dd[0] >> 32)
            print('  net workgroup %d: range %d (slot %d, lap %d) empty for 1 ms at %.3f s into the launch, tail %d head %d, expects tag %d, batches so far %d' % (
                b, base, base & mask, base >> rb, dd[2] / 1e8, dd[1] & 0xFFFFFFFF, dd[1] >> 32, dd[3] & 0xFF, dd[3] >> 16))
            fmt = lambda e_: '(t %d left %d tag %d)' % (e_ & 0xFFFFF, (e_ >> 32) & 0xFFFFFF, e_ >> 60)
            print('     slots as loaded : ' + ' '.join(fmt(int(x)) for x in dd[8:24]))
            print('     slots by atomic : ' + ' '.join(fmt(int(x)) for x in dd[24:40]))
        bs = sorted(wi[n_sel + b][5] for b in range(n_net))
        print('  net workgroups: batches min %d median %d max %d; ranges held at exit below the tail: %s' % (bs[0], bs[len(bs) // 2], bs[-1], sorted((b_, hex(v[1]), v[2]) for b_, v in held.items() if b_ + 16 <= tail)[:12]))
        break

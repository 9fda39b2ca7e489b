"""Offline study on the recorded map-free traces of the bench workload (scripts/dump_traces.py -> gpurun_out/traces_1024_s2000.npz),
on top of scripts/sim_tentative.py's model of the scheduler (tentative replay + running-trace target): DEFER seeds that are likely
to be cut.  A seed whose B^3 block already holds a node of an admitted, unreplayed trace of lower rank (running, paused or finished)
probably sits on a neurite that trace is covering: its own traces would be cut by the replay (or the seed skipped).  Such a seed
is passed over by the admission until it comes within `margin` ranks of the replay frontier (or its block empties).
  python scripts/sim_defer.py [gpurun_out/traces_1024_s2000.npz]"""
import sys
src = open(__file__.replace("sim_defer.py", "sim_tentative.py")).read().split('if __name__ == "__main__":')[0]
src = src.replace("def simulate(window=768,", "def simulate(defer=0, dmargin=16, dblk=8, window=768,")
# admission: scan every unadmitted seed from the frontier on, deferring the ones whose block is taken
src = src.replace('''        s = nxt
        while s < n and s < lim and len(active) + len(paused) + 2 <= window and (not target or len(active) + 2 <= target):
            if state[2 * s] == 0 and s not in held:''', '''        s = frontier if defer else nxt
        while s < n and s < lim and len(active) + len(paused) + 2 <= window and (not target or len(active) + 2 <= target):
            if defer and state[2 * s] == 0 and s not in held and s > frontier + dmargin and dblock(svox[s], dblk) in dblocks:
                ndefer[0] += 1
                s += 1
                continue
            if state[2 * s] == 0 and s not in held:''')
# block bookkeeping: rebuild the set of taken blocks from everything admitted and unreplayed, once per poll
src = src.replace('''        polls += 1
''', '''        polls += 1
        if defer:
            dblocks.clear()
            for g in range(2 * frontier, 2 * min(n, max(nxt, frontier))):
                if state[g] in (1, 2, 3):
                    row = voxl[g]
                    for i in range(min(it[g], Tl[g])):
                        dblocks.add(dblock(row[i], dblk))
''')
src = src.replace('''    den = {}
    it = [0] * (2 * n)''', '''    den = {}
    dblocks = set()
    ndefer = [0]
    it = [0] * (2 * n)''')
src = src.replace("def block_of(v):", '''def dblock(v, B):
    x = v % S; y = (v // S) % S; z = v // (S * S)
    return ((z // B) * 4096 + (y // B)) * 4096 + (x // B)


def block_of(v):''')
src = src.replace("return dict(steps=steps,", "return dict(deferrals=ndefer[0], steps=steps,")
exec(src)
if __name__ == "__main__":
    kw = dict(window=1536, look0=512, look_pct=200, tentative=True, target=200)
    print("base", simulate(**kw), flush=True)
    for blk in (4, 8, 16):
        for margin in (8, 32, 128):
            print("defer block", blk, "margin", margin, simulate(defer=1, dmargin=margin, dblk=blk, **kw), flush=True)

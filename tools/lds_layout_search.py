"""Bank-conflict model of the CU's LDS on gfx950 as measured by tools/ubench_lds.hip / ubench_lds_pat.hip (profiles/r04_ubench_lds*.txt), and a
brute-force search over the lane maps and point offsets of k_implsch4's 36-direction rows under it.  Model: ds_read_b32 and ds_read_b64
are resolved in two half-wave passes (lanes 0-31, 32-63) over 32 banks as wide as the access (4 / 8 bytes); a pass costs the largest number
of distinct addresses on one bank (~1.05 cycles each, 2 at least per instruction).  It reproduces every measured pattern exactly.
Result: with two full points (2 x 18 pairs) in a half-wave no arrangement reads a rotated row without a two-way conflict."""
import itertools, collections
NP, NPAIR = 3, 18
def cost(addrs_bytes, width):
    # two half-wave passes, 32 banks of `width` bytes, pass cost = max distinct addresses per bank
    tot = 0
    for h in (0, 1):
        banks = collections.defaultdict(set)
        for l in range(32*h, 32*h+32):
            a = addrs_bytes[l]
            banks[(a // width) % 32].add(a)
        tot += max(len(v) for v in banks.values())
    return tot
def lane_map(rows):
    # rows: tuple of 4 entries in {'0','1','2','E'}: content of DPP rows 0..3.  returns list of (p, j) per lane
    m = [None]*64
    for r, c in enumerate(rows):
        for i in range(16):
            l = 16*r + i
            if c == 'E':
                m[l] = (i // 2, 16 + (i & 1)) if i < 6 else None   # extras of points 0..2, then shadows
            else:
                m[l] = (int(c), i)
    return m
def addrs(m, bases, shift, half, shadow_of):
    out = []
    for l in range(64):
        pj = m[l] if m[l] is not None else m[shadow_of]
        p, j = pj
        e = ((2*j + shift) % 36 + half) % 36
        out.append(4*(bases[p] + e))
    return out
# access mix per interaction (reads): even b64 shifts with weights, odd shifts as two b32 reads
even_w = {0: 6, 2: 3+1, -2: 3+1, 4: 3+1, -4: 3+1, 6: 1, -6: 1, 8: 1, -8: 1}
odd_w = {1: 3, -1: 3, 3: 3, -3: 3}
def total(rows, bases, shadow_lane):
    m = lane_map(rows)
    t = 0.0
    for s, w in even_w.items():
        t += w * cost(addrs(m, bases, s, 0, shadow_lane), 8)
    for s, w in odd_w.items():
        t += w * (cost(addrs(m, bases, s, 0, shadow_lane), 4) + cost(addrs(m, bases, s, 1, shadow_lane), 4))
    return t
res = []
for rows in set(itertools.permutations(['0','1','2','E'])):
    m = lane_map(rows)
    er = rows.index('E')
    for pad1 in range(0, 10, 2):
        for pad2 in range(0, 10, 2):
            bases = (0, 36+pad1, 72+pad1+pad2)
            for sh in range(64):
                if m[sh] is None: continue
                # shadow must copy a lane; try lanes in the same half as the E row and others
                res.append((total(rows, bases, sh), rows, bases, sh))
res.sort(key=lambda x: x[0])
cur = total(('0','1','2','E'), (0,36,72), 32)
print("current", cur)
seen=set()
for r in res[:400]:
    key=(r[1], r[2])
    if key in seen: continue
    seen.add(key); print(r)
    if len(seen) > 25: break
print("best without padding:")
k=0
for r in res:
    if r[2] == (0,36,72):
        print(r); k+=1
        if k>8: break
print("--- per-instruction, current")
m = lane_map(('0','1','2','E'))
for s in (-4,-2,0,2,4): print("b64", s, cost(addrs(m,(0,36,72),s,0,32),8))
for s in (-3,-1,1,3): print("b32", s, cost(addrs(m,(0,36,72),s,0,32),4), cost(addrs(m,(0,36,72),s,1,32),4))
for b in ((0,37,74),(0,38,76)):
  for s in (-3,-1,1,3): print(b, "b32", s, cost(addrs(m,b,s,0,32),4), cost(addrs(m,b,s,1,32),4))
m2 = lane_map(('0','E','1','2'))
for s in (-4,-2,0,2,4): print("0E12 b64", s, cost(addrs(m2,(0,36,72),s,0,0),8))

"""Offline emulation (numpy, no GPU) of k_emit's prefilter lockstep: issued / useful tests per policy on S2 and S1 clouds of 60 000 atoms.
A wave runs every chunk of a window for as many 8-test groups as its longest lane needs; "useful" counts the records inside each lane's own
window.  Policies: the product's fixed 128-record chunks from the wave's lowest window start; VERDICT r3 item 1c -- end a run when fewer than T
lanes are still busy and restart the next chunk at the lowest unprocessed slot ("adaptive"); other group / chunk sizes.
Result (profiles/r04_emit_experiments.txt section 7): the adaptive restart changes the ratio by under 1 % while adding chunks; 4-test groups
would issue 6 % fewer tests, 256-record chunks 9 % fewer (no LDS for them)."""
import numpy as np, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import synth
SEED = 0xA11CE5EED00
def windows(xyz, kx=4, cutoff=6.5):
    edge = cutoff*(1+1e-6)
    lo = xyz.min(0)
    cx = np.floor((xyz[:,0]-lo[0])*kx/edge).astype(np.int64); cy = np.floor((xyz[:,1]-lo[1])/edge).astype(np.int64); cz = np.floor((xyz[:,2]-lo[2])/edge).astype(np.int64)
    nx, ny, nz = cx.max()+1, cy.max()+1, cz.max()+2
    cell = (cz*ny+cy)*nx+cx
    order = np.argsort(cell, kind='stable')
    cell_s = cell[order]; n=len(cell_s)
    ncells = nx*ny*nz
    start = np.searchsorted(cell_s, np.arange(ncells+1))
    cxs, cys, czs = cx[order], cy[order], cz[order]
    xlo = np.maximum(cxs-kx,0); xhi = np.minimum(cxs+kx, nx-1)
    W = np.zeros((5,n,2), dtype=np.int64)
    W[0,:,0] = np.arange(n)+1; W[0,:,1] = start[(czs*ny+cys)*nx+xhi+1]
    for k,(dy,dz) in enumerate([(1,0),(-1,1),(0,1),(1,1)], start=1):
        yy = cys+dy; zz = czs+dz
        ok = (yy>=0)&(yy<ny)&(zz<nz)
        r = (zz*ny+np.clip(yy,0,ny-1))*nx
        W[k,:,0] = np.where(ok, start[r+xlo], 0); W[k,:,1] = np.where(ok, start[r+xhi+1], 0)
    return W, n
def simulate(W, n, chunk=128, group=8, policy='fixed', T=0):
    issued=0; useful=0; chunks=0
    ntasks=(n+63)//64
    for t in range(ntasks):
        sl = slice(t*64, min(n,(t+1)*64))
        for k in range(5):
            lo = W[k,sl,0].copy(); hi = W[k,sl,1]
            ne = lo<hi
            if not ne.any(): continue
            useful += int((hi-lo)[ne].sum())
            if policy=='fixed':
                L = lo[ne].min(); H = hi[ne].max()
                for cs in range(L, H, chunk):
                    ce=min(cs+chunk,H)
                    j0=np.maximum(lo,cs); j1=np.minimum(hi,ce)
                    ln=np.where(ne&(j1>j0), j1-j0, 0)
                    m=ln.max()
                    if m==0: continue
                    chunks+=1; issued += -(-m//group)*group*64
            else:
                pos = lo.copy()
                while True:
                    act = ne & (pos<hi)
                    if not act.any(): break
                    cs = pos[act].min(); ce = cs+chunk
                    ln = np.where(act, np.minimum(hi,ce)-np.maximum(pos,cs), 0); ln=np.maximum(ln,0)
                    # lanes whose pos > ce have 0
                    chunks+=1
                    # run groups until fewer than T lanes active (but at least one group)
                    g=0
                    while True:
                        g+=1
                        rem = ln - g*group
                        if (rem>0).sum() <= T: break
                    done = np.minimum(ln, g*group)
                    issued += g*group*64
                    pos = np.where(act, np.maximum(pos,cs)+done, pos)
    return issued, useful, chunks, ntasks
for wl,seed in (("s2", SEED+4), ("s1", SEED+3)):
    rec = getattr(synth, f"gen_{wl}")(60000, seed=seed)
    heavy = rec["element"] != b"H"
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], 1)[heavy]
    W,n = windows(xyz)
    for pol,T in (('fixed',0),('adaptive',0),('adaptive',8),('adaptive',16),('adaptive',24)):
        t=time.time(); i,u,c,nt = simulate(W,n,policy=pol,T=T)
        print(wl, pol, T, 'issued/useful %.3f' % (i/64/u*64), 'issued per lane %.1f useful %.1f chunks/task %.2f' % (i/64/nt, u/nt/64*1.0, c/nt), '%.0fs'%(time.time()-t))
print("--- group size / chunk size, fixed policy")
for wl,seed in (("s2", SEED+4), ("s1", SEED+3)):
    rec = getattr(synth, f"gen_{wl}")(60000, seed=seed)
    heavy = rec["element"] != b"H"
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], 1)[heavy]
    W,n = windows(xyz)
    for grp,ch in ((8,128),(4,128),(2,128),(1,128),(8,256),(8,512)):
        i,u,c,nt = simulate(W,n,chunk=ch,group=grp)
        print(wl, 'group', grp, 'chunk', ch, 'issued/useful %.3f' % (i/u), 'chunks/task %.2f' % (c/nt))


def windows_narrow(xyz, kx=4, cutoff=6.5):
    """Per-lane narrowing of the x range of every window by what the home atom's position inside its cell leaves of the cutoff: an atom
    of row (dy, dz) is at least m = [(1 - fy) or fy]^2 + (1 - fz)^2 cell edges away in the y-z plane, so only |dx| <= sqrt(r^2 - m) matters."""
    edge = cutoff*(1+1e-6)
    lo = xyz.min(0)
    fx = (xyz[:,0]-lo[0])*kx/edge; fyv = (xyz[:,1]-lo[1])/edge; fzv = (xyz[:,2]-lo[2])/edge
    cx = np.floor(fx).astype(np.int64); cy = np.floor(fyv).astype(np.int64); cz = np.floor(fzv).astype(np.int64)
    nx, ny, nz = cx.max()+1, cy.max()+1, cz.max()+2
    cell = (cz*ny+cy)*nx+cx
    order = np.argsort(cell, kind='stable')
    cell_s = cell[order]; n=len(cell_s)
    start = np.searchsorted(cell_s, np.arange(nx*ny*nz+1))
    cxs, cys, czs = cx[order], cy[order], cz[order]
    fxs = fx[order]; fy = (fyv-cy)[order]; fz = (fzv-cz)[order]
    W = np.zeros((5,n,2), dtype=np.int64)
    xhi0 = np.minimum(cxs+kx, nx-1)
    W[0,:,0] = np.arange(n)+1; W[0,:,1] = start[(czs*ny+cys)*nx+xhi0+1]
    for k,(dy,dz) in enumerate([(1,0),(-1,1),(0,1),(1,1)], start=1):
        m = np.zeros(n)
        if dy == 1: m += (1-fy)**2
        if dy == -1: m += fy**2
        if dz == 1: m += (1-fz)**2
        half = np.sqrt(np.maximum(1.0 - m, 0.0)) * kx * (1+1e-9)   # in x cells; m >= 1 -> nothing can be in range
        xl = np.maximum(np.floor(fxs - half).astype(np.int64), 0); xh = np.minimum(np.floor(fxs + half).astype(np.int64), nx-1)
        yy = cys+dy; zz = czs+dz
        ok = (yy>=0)&(yy<ny)&(zz<nz)&(m < 1.0)
        r = (zz*ny+np.clip(yy,0,ny-1))*nx
        W[k,:,0] = np.where(ok, start[r+xl], 0); W[k,:,1] = np.where(ok, start[r+np.maximum(xh,xl)+1], 0)
    return W, n


print("--- per-lane narrowing of the x ranges (fixed chunks, 8-test groups)")
for wl,seed in (("s2", SEED+4), ("s1", SEED+3)):
    rec = getattr(synth, f"gen_{wl}")(60000, seed=seed)
    heavy = rec["element"] != b"H"
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], 1)[heavy]
    for name, fn in (("product", windows), ("narrowed", windows_narrow)):
        W,n = fn(xyz)
        i,u,c,nt = simulate(W,n)
        print(wl, name, 'issued per lane %.1f  useful per lane %.1f  issued/useful %.3f  chunks/task %.2f' % (i/64/nt, u/64/nt, i/u, c/nt))

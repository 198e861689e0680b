// SAP neighbour sum on the cell list (reference src/sap.rs:155-204).  Included by kernels.hip inside namespace arp.
//
// For every side-chain atom x: the f32 sum, over the side-chain atoms y within sap_radius of x (x itself included, inclusive test in
// f64 as rstar's locate_within_distance), of weight(y) = hydrophobicity(resn(y)) * clamp(sasa(y) / max_sc_asa(resn(y)), 0, 1).  Same
// access pattern as the contact search, different reduction: the grid is built over the side-chain atoms only (everything else
// carries the "not in the grid" attribute bit), one thread per home slot walks the shell of cells within one cutoff (nine contiguous slot windows:
// cells are x-major, 2 kx + 1 of them per window) and accumulates in slot order -- the ordered grid makes that order, and so the f32 sum, reproducible.
__global__ __launch_bounds__(256) void k_neighbor_sum(const GridParams *gp, const uint32_t *cell_start, Sorted so, double r2, const float *weight, float *out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nx = gp->nx, ny = gp->ny, nzt = gp->nzt, kx = gp->kx;
    if (p >= gp->n_heavy) return;
    const Fat a = so.fat[p];
    const uint32_t c = a.cell, cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    const uint32_t xlo = cx > kx ? cx - kx : 0u, xhi = min(cx + kx, nx - 1);
    float acc = 0.0f;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++) {
            const int zz = (int)cz + dz, yy = (int)cy + dy;
            if (zz < 0 || zz >= (int)nzt || yy < 0 || yy >= (int)ny) continue;
            const uint32_t r = ((uint32_t)zz * ny + (uint32_t)yy) * nx;
            for (uint32_t q = cell_start[r + xlo], q1 = cell_start[r + xhi + 1]; q < q1; q++) {
                const Fat b = so.fat[q];
                if ((a.crm >> 16) == (b.crm >> 16) && sq_dist(a.x, a.y, a.z, b.x, b.y, b.z) <= r2) acc += weight[b.orig];
            }
        }
    out[a.orig] = acc;
}

void launch_neighbor_sum(const DevAtoms &in, const Workspace &ws, double radius, double r2, const float *weight, float *out, hipStream_t st) {
    launch_grid(in, ws, st, nullptr, radius, /* ordered: slots follow the atom index inside a cell */ true);
    if (in.n) hipLaunchKernelGGL(k_neighbor_sum, dim3((in.n + 255u) / 256u), dim3(256), 0, st, (const GridParams *)ws.grid, (const uint32_t *)ws.cell_start, ws.sorted, r2,
                                 weight, out);
}

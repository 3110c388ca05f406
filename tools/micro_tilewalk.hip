// Experiment (round 2): what does the ORDER in which the lower-triangle tiles of the column-major covariance are read
// and rewritten cost?  Reads every element of the tiles on/below the diagonal of an n x n fp32 matrix (ld = npad) once
// and writes it back (x -> x * 1.0001f), nothing else -- the memory side of the down-date without panels or MFMAs.
//
//   mode 0  tile at a time, tiles handed out round-robin (the order of the first-round kernel's micro-benchmark)
//   mode 1  COLUMN STRIPS: a workgroup walks down one 128-column band, R0 = 128 rows per step
//   mode 2  the same with 256 rows per step (two vertically adjacent tiles: 1 KB contiguous per column)
//   mode 3  the same with 512 rows per step
//   mode 4  ROW STRIPS: a workgroup walks along one 128-row band, 128 columns per step (the tile order of a row walk)
// each with plain and with non-temporal (aux = 2) buffer accesses, 16 bytes per lane.
// Work is split into contiguous segments of equal size, one per persistent workgroup (grid = 2 x CUs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Seg { int col, row0, nsteps, rows; };   // a run of `nsteps` blocks of `rows` x 128 starting at (row0, col)

template <int AUX>
__global__ __launch_bounds__(512) void walk(float* P, int ld, const Seg* segs, const int* seg_first, int rowstrip) {
    const int tid = threadIdx.x;
    // thread -> (column c of the 128, 16-byte piece q of a 128-row run): 32 pieces per column per 128 rows
    for (int s = seg_first[blockIdx.x]; s < seg_first[blockIdx.x + 1]; ++s) {
        const Seg sg = segs[s];
        for (int st = 0; st < sg.nsteps; ++st) {
            const int r0 = rowstrip ? sg.row0 : sg.row0 + st * sg.rows;
            const int c0 = rowstrip ? sg.col + st * 128 : sg.col;
            // rows x 128 columns = rows*128/4 float4; 512 threads
            const int per = sg.rows / 4;                 // float4 per column
            const int total = per * 128;
            for (int base = 0; base < total; base += 512 * 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 512 + tid;
                    const int c = idx / per, q = idx - c * per;
                    const float* a = P + (size_t)(c0 + c) * ld + r0 + 4 * q;
                    if (idx < total) v[u] = AUX ? __builtin_nontemporal_load((const f32x4*)a) : *(const f32x4*)a;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 512 + tid;
                    const int c = idx / per, q = idx - c * per;
                    float* a = P + (size_t)(c0 + c) * ld + r0 + 4 * q;
                    if (idx < total) {
                        v[u] *= 1.0001f;
                        if (AUX) __builtin_nontemporal_store(v[u], (f32x4*)a); else *(f32x4*)a = v[u];
                    }
                }
            }
        }
    }
}

// tile-major storage: tile k of the walk is the contiguous 64 KiB block k (band-major order = one linear stream)
template <int AUX>
__global__ __launch_bounds__(512) void walk_tm(float* P, int ntiles) {
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        f32x4* base = (f32x4*)(P + (size_t)t * 16384);
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = AUX ? __builtin_nontemporal_load(base + u * 512 + tid) : base[u * 512 + tid];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] *= 1.0001f;
            if (AUX) __builtin_nontemporal_store(v[u], base + u * 512 + tid); else base[u * 512 + tid] = v[u];
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20003;
    const int T = (n + 127) / 128;
    const int ld = T * 128 + (argc > 2 ? atoi(argv[2]) : 0);            // experiment: extra leading-dimension padding (floats)
    const int only = argc > 3 ? atoi(argv[3]) : -1;
    float* P;
    hipMalloc(&P, (size_t)ld * ld * 4);
    hipMemset(P, 0, (size_t)ld * ld * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int nwg = 2 * prop.multiProcessorCount;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const double bytes = 2.0 * (double)T * (T + 1) / 2 * 65536.0;
    {   // mode 9 / 10: tile-major storage, band-major order; one workgroup per tile / persistent grid
        const int ntiles = T * (T + 1) / 2;
        for (int pers = 0; pers < 2; ++pers)
            for (int aux = 0; aux < 2; ++aux) {
                const int grid = pers ? nwg : ntiles;
                for (int rep = 0; rep < 3; ++rep) { if (aux) walk_tm<2><<<grid, 512>>>(P, ntiles); else walk_tm<0><<<grid, 512>>>(P, ntiles); }
                hipEventRecord(a);
                for (int rep = 0; rep < 10; ++rep) { if (aux) walk_tm<2><<<grid, 512>>>(P, ntiles); else walk_tm<0><<<grid, 512>>>(P, ntiles); }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("n %d tile-major band-major %s %s: %.4f ms per pass, %.2f TB/s\n", n, pers ? "persistent   " : "1 WG per tile", aux ? "nt   " : "plain", ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
            }
    }
    for (int mode = 0; mode < 9; ++mode) {
        if (only >= 0 && mode != only && mode != 0 && !(only == 7 && mode == 8)) continue;
        const int rows = mode == 2 ? 256 : mode == 3 ? 512 : 128;
        std::vector<Seg> units;      // unit blocks in walk order
        if (mode == 0) {
            for (int I = 0; I < T; ++I) for (int J = 0; J <= I; ++J) units.push_back({J * 128, I * 128, 1, 128});
        } else if (mode == 7) {          // mode 5's order, ONE workgroup per tile (hardware dispatch instead of a persistent grid)
            for (int J = 0; J < T; ++J) for (int I = J; I < T; ++I) units.push_back({J * 128, I * 128, 1, 128});
        } else if (mode == 8) {          // mode 0's order, one workgroup per tile
            for (int I = 0; I < T; ++I) for (int J = 0; J <= I; ++J) units.push_back({J * 128, I * 128, 1, 128});
        } else if (mode == 5) {          // band-major: the tiles of one 128-column band are in flight together
            for (int J = 0; J < T; ++J) for (int I = J; I < T; ++I) units.push_back({J * 128, I * 128, 1, 128});
        } else if (mode == 6) {          // band-major, rows dealt to the 8 XCDs by I % 8 (workgroup b sits on XCD b % 8)
            std::vector<std::vector<Seg>> x(8);
            for (int J = 0; J < T; ++J) for (int I = J; I < T; ++I) x[I % 8].push_back({J * 128, I * 128, 1, 128});
            size_t L = 0; for (auto& v : x) L = v.size() > L ? v.size() : L;
            for (size_t i = 0; i < L; ++i) for (int k = 0; k < 8; ++k) if (i < x[k].size()) units.push_back(x[k][i]); else units.push_back({0, 0, 0, 128});
        } else if (mode == 4) {
            for (int I = 0; I < T; ++I) units.push_back({0, I * 128, I + 1, 128});
        } else {
            for (int J = 0; J < T; ++J) {
                const int nrows = (T - J) * 128;        // from the diagonal tile down
                const int full = nrows / rows;
                if (full) units.push_back({J * 128, J * 128, full, rows});
                if (nrows % rows) units.push_back({J * 128, J * 128 + full * rows, (nrows % rows) / 128, 128});
            }
        }
        // cut the walk into nwg segments of equal tile count (strips are split where needed)
        std::vector<Seg> segs;
        std::vector<int> first(nwg + 1, 0);
        const double total_tiles = (double)T * (T + 1) / 2;
        double acc = 0;
        int w = 0;
        int grid = nwg;
        if (mode >= 7) {
            grid = (int)units.size();
            segs = units;
            first.resize(grid + 1);
            for (int k = 0; k <= grid; ++k) first[k] = k;
        } else if (mode == 0 || mode >= 5) {            // round robin
            std::vector<std::vector<Seg>> per(nwg);
            for (size_t i = 0; i < units.size(); ++i) per[i % nwg].push_back(units[i]);
            for (int k = 0; k < nwg; ++k) { first[k] = (int)segs.size(); for (auto& s : per[k]) segs.push_back(s); }
            first[nwg] = (int)segs.size();
        } else {
            first[0] = 0;
            for (auto u : units) {
                while (u.nsteps > 0) {
                    const double tiles_per_step = u.rows / 128.0;
                    const double room = total_tiles * (w + 1) / nwg - acc;
                    int take = (int)(room / tiles_per_step + 0.5);
                    if (take < 1) take = 1;
                    if (take > u.nsteps) take = u.nsteps;
                    segs.push_back({u.col, u.row0, take, u.rows});
                    acc += take * tiles_per_step;
                    if (mode == 4) u.col += take * 128; else u.row0 += take * u.rows;
                    u.nsteps -= take;
                    while (w + 1 < nwg && acc >= total_tiles * (w + 1) / nwg - 1e-9) first[++w] = (int)segs.size();
                }
            }
            for (int k = w + 1; k <= nwg; ++k) first[k] = (int)segs.size();
        }
        Seg* dsegs; int* dfirst;
        hipMalloc(&dsegs, segs.size() * sizeof(Seg));
        hipMalloc(&dfirst, first.size() * sizeof(int));
        hipMemcpy(dsegs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice);
        hipMemcpy(dfirst, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice);
        for (int aux = 0; aux < 2; ++aux) {
            for (int rep = 0; rep < 3; ++rep) {
                if (aux) walk<2><<<grid, 512>>>(P, ld, dsegs, dfirst, mode == 4); else walk<0><<<grid, 512>>>(P, ld, dsegs, dfirst, mode == 4);
            }
            hipEventRecord(a);
            for (int rep = 0; rep < 10; ++rep) {
                if (aux) walk<2><<<grid, 512>>>(P, ld, dsegs, dfirst, mode == 4); else walk<0><<<grid, 512>>>(P, ld, dsegs, dfirst, mode == 4);
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("n %d ld %d mode %d (%s) %s: %.4f ms per pass, %.2f TB/s   [%zu segments]\n", n, ld, mode,
                   mode == 0 ? "tiles round-robin" : mode == 5 ? "tiles band-major rr" : mode == 7 ? "band-major, 1 WG per tile" : mode == 8 ? "row-major, 1 WG per tile" : mode == 6 ? "band-major, XCD = I%8" : mode == 4 ? "row strips" : mode == 1 ? "column strips x128" : mode == 2 ? "column strips x256" : "column strips x512",
                   aux ? "nt   " : "plain", ms / 10, bytes / (ms / 10 * 1e-3) / 1e12, segs.size());
        }
        hipFree(dsegs); hipFree(dfirst);
    }
    return 0;
}

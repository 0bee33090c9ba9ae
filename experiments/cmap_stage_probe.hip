// cmap_stage_probe.hip -- developer probe (round 5; not part of the library).  Where does the contact stage's CSR / letter-sum kernel
// (k_cmap_fill_rows) spend its time?  The kernel is compiled here with MDF_FILL_STAMPS: thread 0 of every block stamps the 100 MHz realtime
// counter and the shader clock at five points, and wave 0 reports its loop trips.
//   PROBE_NO_LS=1        the kernel without the letter sums (no LDS adds)
//   MDFRI_FILL_NOCSR=1   the kernel without the CSR stores (results incomplete: timing only)
// Findings (128 x L512 = one 65 536-row chunk, profiles/r05_cmap_fill_probe.txt): with one lane per ROW (64 rows per wave, one wave per
// SIMD) the kernel's 32 us were the latency of the wave holding the densest row -- 25 trips on average, 67 at most, ~1 100 shader cycles
// per trip; with eight lanes per row (8 rows per wave, 8 waves per SIMD: the shipped form) a trip costs a SIMD ~115 cycles of issue
// (29 instructions), the loop 8 us on average and 16 us in the densest block, the kernel 22 us; neither the CSR stores nor the LDS adds
// are what a trip costs (7.2 / 7.6 us without either).
//   build:  make -C experiments bin/cmap_stage_probe      run:  experiments/bin/cmap_stage_probe [proteins] [length]
#define MDF_FILL_STAMPS 1
#include "../metagenomic-deepfri_amd/csrc/cmap.hip"

#include <random>
#include <string>

using namespace mdf;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("%s -> %s\n", #x, hipGetErrorString(e));                            \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 128, L = argc > 2 ? atoi(argv[2]) : 512;
    std::mt19937 rng(3);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> xyz((size_t)B * L * 3);
    for (int p = 0; p < B; ++p) {
        float x = 0, y = 0, z = 0;
        for (int i = 0; i < L; ++i) {
            float dx = g(rng), dy = g(rng), dz = g(rng);
            const float n = 3.8f / std::sqrt(dx * dx + dy * dy + dz * dz);
            x += dx * n; y += dy * n; z += dz * n;
            float *o = &xyz[((size_t)p * L + i) * 3];
            o[0] = std::round(x * 1000.f) / 1000.f; o[1] = std::round(y * 1000.f) / 1000.f; o[2] = std::round(z * 1000.f) / 1000.f;
        }
    }
    std::string aln((size_t)B * L, 'A');
    std::vector<uint8_t> letters;
    std::vector<int32_t> Lq(B, L), row_off(B + 1), coord_off(B + 1), aln_off(B + 1);
    const int64_t R = mdf_layout_rows(Lq.data(), B, row_off.data());
    letters.resize((size_t)R);
    for (auto &c : letters) c = (uint8_t)(rng() % 20 + 1);
    for (int p = 0; p <= B; ++p) coord_off[p] = aln_off[p] = p * L;
    float *d_xyz, *d_val, *d_ls;
    char *d_q, *d_t;
    int32_t *d_lq, *d_ro, *d_co, *d_ao, *d_rowptr, *d_col, *d_status;
    uint8_t *d_let;
    void *d_ws;
    const int64_t cap = R * 64;
    const size_t wsb = mdf_cmap_workspace_bytes(B, R, L);
    CK(hipMalloc(&d_xyz, xyz.size() * 4)); CK(hipMalloc(&d_q, aln.size())); CK(hipMalloc(&d_t, aln.size()));
    CK(hipMalloc(&d_lq, (B + 1) * 4)); CK(hipMalloc(&d_ro, (B + 1) * 4)); CK(hipMalloc(&d_co, (B + 1) * 4)); CK(hipMalloc(&d_ao, (B + 1) * 4));
    CK(hipMalloc(&d_rowptr, (R + 1) * 4)); CK(hipMalloc(&d_col, cap * 4)); CK(hipMalloc(&d_val, cap * 4)); CK(hipMalloc(&d_status, 16));
    CK(hipMalloc(&d_let, R)); CK(hipMalloc(&d_ls, (size_t)R * 128)); CK(hipMalloc(&d_ws, wsb));
    CK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_q, aln.data(), aln.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_t, aln.data(), aln.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lq, Lq.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ro, row_off.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_co, coord_off.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ao, aln_off.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_let, letters.data(), R, hipMemcpyHostToDevice));
    CK(hipMemset(d_status, 0, 16));
    const int blocks = (int)(R / FILL_ROWS);
    unsigned long long *d_stamps;
    CK(hipMalloc(&d_stamps, (size_t)blocks * 12 * 8));
    CK(hipMemset(d_stamps, 0, (size_t)blocks * 12 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_fill_stamps), &d_stamps, sizeof(d_stamps)));
    const int rblocks = (int)(R / 32);
    unsigned long long *d_rstamps;
    CK(hipMalloc(&d_rstamps, (size_t)rblocks * 8 * 8));
    CK(hipMemset(d_rstamps, 0, (size_t)rblocks * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamps), &d_rstamps, sizeof(d_rstamps)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&]() {
        return mdf_cmap_csr_dev(d_xyz, d_co, d_q, d_t, d_ao, d_lq, d_ro, B, R, L, 6.0, 2, d_rowptr, d_col, d_val, cap, d_status, getenv("PROBE_NO_LS") ? nullptr : d_let, getenv("PROBE_NO_LS") ? nullptr : d_ls, d_ws, wsb, nullptr);
    };
    for (int i = 0; i < 3; ++i)
        if (run()) { printf("cmap_csr_dev failed: %s\n", mdf_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)blocks * 12);
    CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> rp((size_t)R + 1);
    CK(hipMemcpy(rp.data(), d_rowptr, rp.size() * 4, hipMemcpyDeviceToHost));
    printf("%d proteins x L=%d, R=%lld rows, %d entries (%.1f per row); contact stage (4 launches) %.1f us per call\n", B, L, (long long)R, rp[R], (double)rp[R] / (B * L), ms * 1e3 / reps);
    const char *names[4] = {"prologue + staging", "barrier", "entry loop", "letter sums out"};
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < blocks; ++b) { first = std::min(first, st[12ull * b]); last = std::max(last, st[12ull * b + 4]); }
    printf("k_cmap_fill_rows: first block start .. last block end %.2f us\n", (last - first) * 0.01);
    for (int ph = 0; ph < 4; ++ph) {
        double ns = 0, cyc = 0, mx = 0;
        for (int b = 0; b < blocks; ++b) {
            const double d = (double)(st[12ull * b + ph + 1] - st[12ull * b + ph]) * 10.0;
            ns += d; mx = std::max(mx, d);
            cyc += (double)(st[12ull * b + 5 + ph + 1] - st[12ull * b + 5 + ph]);
        }
        printf("  %-20s mean %8.0f ns  max %8.0f ns  mean %8.0f shader cycles (%.2f GHz)\n", names[ph], ns / blocks, mx, cyc / blocks, cyc / ns);
    }
    double trips = 0, ent = 0, start = 0;
    int tmax = 0;
    for (int b = 0; b < blocks; ++b) { trips += st[12ull * b + 10]; ent += st[12ull * b + 11]; tmax = std::max(tmax, (int)st[12ull * b + 10]); start += (double)(st[12ull * b] - first) * 10.0; }
    printf("  loop trips per wave: mean %.1f max %d; entries per wave %.0f; mean block start %.0f ns after the first\n", trips / blocks, tmax, ent / blocks, start / blocks);
    {
        std::vector<unsigned long long> rs((size_t)rblocks * 8);
        CK(hipMemcpy(rs.data(), d_rstamps, rs.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long f = ~0ull, l = 0;
        for (int b = 0; b < rblocks; ++b) { f = std::min(f, rs[8ull * b]); l = std::max(l, rs[8ull * b + 5]); }
        printf("k_cmap_rows<COUNT>: first block start .. last block end %.2f us (L <= 1 024: one column tile)\n", (l - f) * 0.01);
        const char *rn[5] = {"prologue (rows)", "column staging", "barrier", "distance loop", "counts out"};
        for (int ph = 0; ph < 5; ++ph) {
            double ns = 0, mx = 0;
            for (int b = 0; b < rblocks; ++b) { const double d = (double)(rs[8ull * b + ph + 1] - rs[8ull * b + ph]) * 10.0; ns += d; mx = std::max(mx, d); }
            printf("  %-20s mean %8.0f ns  max %8.0f ns\n", rn[ph], ns / rblocks, mx);
        }
        double start = 0;
        for (int b = 0; b < rblocks; ++b) start += (double)(rs[8ull * b] - f) * 10.0;
        printf("  mean block start %.0f ns after the first\n", start / rblocks);
    }
    return 0;
}

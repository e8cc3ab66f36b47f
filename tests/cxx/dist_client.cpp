// A C++ launcher of the sharded VO job (include/voldor_hip.h section D), built with plain g++ and linked with -lvoldor_hip: no Python and no
// torch anywhere in the process.  usage: dist_client <rank> <world> <rendezvous file> [device]
// Every rank runs its share of SEQ synthetic sequences (a camera moving towards a fronto-parallel plane at a sequence-specific speed) through
// vk_voldor_sharded and checks that it ends up with EVERY sequence's record, equal to the record of a one-at-a-time vk_voldor_device call
// for sequences it can recompute itself.  Prints "DIST CLIENT OK <rank>".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "voldor_hip.h"

static const int W = 96, H = 72, NF = 2, SEQ = 5;
static std::vector<float> make_flows(int seq) {
    const float fx = 48.f, fy = 48.f, cx = 48.f, cy = 36.f, Z = 8.f, tz = 0.2f + 0.05f * seq;
    std::vector<float> flows((size_t)NF * W * H * 2);
    for (int f = 0; f < NF; f++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float Zf = Z - tz * f, X = (x - cx) / fx * Zf, Y = (y - cy) / fy * Zf;
                float* o = &flows[(((size_t)f * H + y) * W + x) * 2];
                o[0] = fx * X / (Zf - tz) + cx - x;
                o[1] = fy * Y / (Zf - tz) + cy - y;
            }
    return flows;
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: dist_client rank world file [device]\n"); return 64; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), dev = argc > 4 ? atoi(argv[4]) : 0;
    if (vk_set_device(dev) != 0) return 1;
    if (vk_dist_init_file(rank, world, argv[3], 60) != 0) return 2;
    if (vk_dist_rank() != rank || vk_dist_world() != world) return 3;
    const int len = 1 + 42 * NF;
    // contiguous, balanced shards (the first SEQ % world ranks get one more): the rule of voldor_amd/dist.py
    auto shard_lo = [&](int r) { const int base = SEQ / world, rem = SEQ % world; return r * base + (r < rem ? r : rem); };
    auto shard_n = [&](int r) { return SEQ / world + (r < SEQ % world ? 1 : 0); };
    const int steps = (SEQ + world - 1) / world;
    std::vector<std::vector<float>> result(SEQ);
    std::vector<float> records((size_t)world * len), poses(NF * 6), covar(NF * 36);
    for (int s = 0; s < steps; s++) {
        const bool have = s < shard_n(rank);
        std::vector<float> flows;
        if (have) flows = make_flows(shard_lo(rank) + s);
        int n = 0;
        vk_set_rand_epoch(0);
        if (vk_voldor_sharded(have ? flows.data() : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 48.f, 48.f, 48.f, 36.f, 0.f, NF, 0, W, H,
                              "--silent --max_iters 3", &n, poses.data(), covar.data(), nullptr, nullptr, records.data()) != 0) return 4;
        for (int r = 0; r < world; r++) {
            const float* rec = &records[(size_t)r * len];
            if (s < shard_n(r)) { if (rec[0] != (float)NF) { printf("rank %d step %d: record of rank %d says %f frames\n", rank, s, r, rec[0]); return 5; }
                                  result[shard_lo(r) + s].assign(rec, rec + len); }
            else if (rec[0] != -1.f) return 6;  // an empty slot must be marked
        }
    }
    for (int q = 0; q < SEQ; q++) {  // every sequence arrived; and it is what the plain window call gives
        if ((int)result[q].size() != len) return 7;
        std::vector<float> flows = make_flows(q);
        int n = 0;
        vk_set_rand_epoch(0);
        if (vk_voldor_device(flows.data(), nullptr, nullptr, nullptr, nullptr, nullptr, 48.f, 48.f, 48.f, 36.f, 0.f, NF, 0, W, H, "--silent --max_iters 3", &n,
                             poses.data(), covar.data(), nullptr, nullptr) != 0 || n != NF) return 8;
        if (memcmp(&result[q][1], poses.data(), sizeof(float) * 6 * NF) != 0 || memcmp(&result[q][1 + 6 * NF], covar.data(), sizeof(float) * 36 * NF) != 0) {
            printf("rank %d: sequence %d differs from the one-at-a-time window\n", rank, q);
            return 9;
        }
    }
    double t = 1.0 + rank;
    if (vk_dist_allreduce_max(&t) != 0 || t != (double)world) return 10;
    if (vk_dist_barrier() != 0) return 11;
    vk_dist_finalize();
    printf("DIST CLIENT OK %d\n", rank);
    return 0;
}

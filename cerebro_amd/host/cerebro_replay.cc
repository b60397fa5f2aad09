// cerebro_replay -- offline replay harness of the loop-candidate producer (BASELINE configs 1 and 5: "harness ready,
// data absent": feed it the descriptors of an EuRoC run and a tick schedule and it reproduces foundLoops).
//
//   cerebro_replay [--devices a,b,..] <stream.bin> <out.json> [device]
// stream.bin (little endian): "CRBR" u32 version=1, u32 D, u64 N, u64 n_ticks, then N x {u32 sec, u32 nsec},
// N x D float64 descriptors (the .srv wire type), n_ticks x i64 l (value of wholeImageComputedList_size() at each
// iteration of the dot-product thread; rows < l are appended before the tick).  Optional trailers, any order, so that the dump's
// global_a / global_b are the reference's data_map indices (src/Cerebro.cpp:1142-1143) and a genuine loopcandidates_liverun.json
// compares without field exceptions:  "FRMS" u64 F, F x {u32 sec, u32 nsec} = the stamp of EVERY camera frame (what DataManager's
// data_map holds);  "FIDX" u64 N, N x i64 = row -> data_map index, given directly.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cerebro_host.h"
#include "state_json.h"

// --parse-only <state.json> <out.bin>  : no GPU; dumps {u32 D, u64 n, n x u64 stampNSec, n*D x f64} (parser check)
// --state <state.json> <out.json> [device] : cold start from a checkpoint, then tick every 3 rows from l = 56
static int parse_only(const char *in, const char *outp)
{
    cerebro_hip::StateDescriptors sd;
    if (!cerebro_hip::load_state_json(in, sd)) { std::fprintf(stderr, "parse error: %s\n", sd.error.c_str()); return 6; }
    FILE *o = std::fopen(outp, "wb");
    if (!o) { std::perror(outp); return 2; }
    const uint32_t D = (uint32_t)sd.D;
    const uint64_t n = sd.stampNSec.size();
    std::fwrite(&D, 4, 1, o);
    std::fwrite(&n, 8, 1, o);
    std::fwrite(sd.stampNSec.data(), 8, n, o);
    std::fwrite(sd.desc.data(), 8, sd.desc.size(), o);
    std::fclose(o);
    std::fprintf(stderr, "cerebro_replay: %lld nodes, %llu descriptors of %u\n", (long long)sd.n_nodes, (unsigned long long)n, D);
    return 0;
}

// --devices a,b,c (first option): run the loop detector over several GPUs of this node from this one process
// (chip_create_multi; a device named twice = the device-copy exchange, e.g. --devices 0,0,0,0 on a 1-GPU machine).
static std::vector<int> g_devices;

static cerebro_hip::Cerebro *make_cerebro(int D, int device, int64_t hint)
{
    return g_devices.empty() ? new cerebro_hip::Cerebro(D, device, hint) : new cerebro_hip::Cerebro(D, g_devices, hint);
}

static int from_state(const char *in, const char *outp, int device)
{
    cerebro_hip::StateDescriptors sd;
    if (!cerebro_hip::load_state_json(in, sd)) { std::fprintf(stderr, "parse error: %s\n", sd.error.c_str()); return 6; }
    std::unique_ptr<cerebro_hip::Cerebro> cer_p(make_cerebro(sd.D, device, (int64_t)sd.stampNSec.size()));
    cerebro_hip::Cerebro &cer = *cer_p;
    if (!cer.ok()) { std::fprintf(stderr, "chip_create failed: %s\n", chip_strerror(cer.last_status())); return 3; }
    const int64_t n = cer.loadStateFromDisk(in);
    if (n < 0) { std::fprintf(stderr, "loadStateFromDisk: %s\n", cer.last_error().c_str()); return 4; }
    for (int64_t l = 56; l <= n; l += 3) {   // SURVEY 8d replay schedule: l advances by 3 from the first productive tick
        cer.descrip_N__dot__descrip_0_N_once(l);
        if (cer.last_status() != CHIP_OK) { std::fprintf(stderr, "tick failed: %s\n", chip_strerror(cer.last_status())); return 5; }
    }
    FILE *o = std::fopen(outp, "w");
    if (!o) { std::perror(outp); return 2; }
    const std::string js = cer.foundLoops_as_JSON();
    std::fwrite(js.data(), 1, js.size(), o);
    std::fclose(o);
    std::fprintf(stderr, "cerebro_replay: %lld descriptors from state.json, %d loop candidates\n", (long long)n, cer.foundLoops_count());
    return 0;
}

// --compare <reference.json> <candidate.json> : no GPU.  Diffs two loop-candidate dumps in the format the reference's node
// writes at shutdown (loopcandidates_liverun.json = Cerebro::foundLoops_as_JSON().dump(4), src/cerebro_node.cpp:769-770, keys
// of src/Cerebro.cpp:1149-1159) -- e.g. a recorded EuRoC run of the reference against `cerebro_replay <stream> <out.json>` fed
// the same descriptors and tick schedule (BASELINE configs 1 and 5).  Candidates are matched IN ORDER on the four time stamps
// (sec/nsec of t_curr and t_prev): that is the selection.  global_a / global_b (the reference: index into DataManager's data_map,
// src/Cerebro.cpp:1142-1143) are compared too -- a replay that was given the frames (stream trailers FRMS / FIDX, or a state.json
// cold start) emits the same indices; `--compare-selection` skips them for a replay that was not.  Scores are compared
// numerically and the largest difference is reported (Eigen's summation order differs from the fixed tree in the last bits).
// Prints one JSON object; exit status 0 iff the two dumps agree.
struct LoopRec { uint64_t sa = 0, na = 0, sb = 0, nb = 0; double score = 0.0; int have = 0; long long ga = -1, gb = -1; };

static bool parse_loop_dump(const char *path, std::vector<LoopRec> &out, std::string &err)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::string t;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) t.append(buf, n);
    std::fclose(f);
    const char *p = t.c_str(), *e = p + t.size();
    auto ws = [&] { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; };
    ws();
    if (p < e && std::strncmp(p, "null", 4) == 0) return true;   // nlohmann dumps an empty (never pushed-to) json as null
    if (p >= e || *p != '[') { err = "expected a JSON array"; return false; }
    p++;
    for (;;) {
        ws();
        if (p < e && *p == ']') return true;
        if (p >= e || *p != '{') { err = "expected an object"; return false; }
        p++;
        LoopRec r;
        for (;;) {
            ws();
            if (p < e && *p == '}') { p++; break; }
            if (p >= e || *p != '"') { err = "expected a key"; return false; }
            const char *k = ++p;
            while (p < e && *p != '"') p++;
            if (p >= e) { err = "unterminated key"; return false; }
            const std::string key(k, p);
            p++;
            ws();
            if (p >= e || *p != ':') { err = "expected ':'"; return false; }
            p++;
            ws();
            char *q = nullptr;
            if (key == "score") { r.score = std::strtod(p, &q); r.have |= 16; }
            else if (key == "time_sec_a") { r.sa = std::strtoull(p, &q, 10); r.have |= 1; }
            else if (key == "time_nsec_a") { r.na = std::strtoull(p, &q, 10); r.have |= 2; }
            else if (key == "time_sec_b") { r.sb = std::strtoull(p, &q, 10); r.have |= 4; }
            else if (key == "time_nsec_b") { r.nb = std::strtoull(p, &q, 10); r.have |= 8; }
            else if (key == "global_a") { r.ga = std::strtoll(p, &q, 10); r.have |= 32; }
            else if (key == "global_b") { r.gb = std::strtoll(p, &q, 10); r.have |= 64; }
            else (void)std::strtod(p, &q);   // time_double_*: derived from sec / nsec
            if (!q || q == p) { err = "expected a number for key " + key; return false; }
            p = q;
            ws();
            if (p < e && *p == ',') p++;
        }
        if ((r.have & 31) != 31) { err = "candidate without time_sec/nsec_a/b + score"; return false; }
        out.push_back(r);
        ws();
        if (p < e && *p == ',') p++;
    }
}

static int compare_dumps(const char *ref_path, const char *cand_path, bool with_global)
{
    std::vector<LoopRec> a, b;
    std::string err;
    if (!parse_loop_dump(ref_path, a, err)) { std::fprintf(stderr, "%s: %s\n", ref_path, err.c_str()); return 6; }
    if (!parse_loop_dump(cand_path, b, err)) { std::fprintf(stderr, "%s: %s\n", cand_path, err.c_str()); return 6; }
    size_t n = a.size() < b.size() ? a.size() : b.size(), first = n;
    double max_d = 0.0;
    for (size_t i = 0; i < n; i++) {
        if (a[i].sa != b[i].sa || a[i].na != b[i].na || a[i].sb != b[i].sb || a[i].nb != b[i].nb) { first = i; break; }
        const double d = a[i].score > b[i].score ? a[i].score - b[i].score : b[i].score - a[i].score;
        if (d > max_d) max_d = d;
    }
    const bool same_sel = first == n && a.size() == b.size();
    size_t global_mismatch = 0, global_compared = 0;
    if (with_global)
        for (size_t i = 0; i < first; i++) {
            if ((a[i].have & 96) != 96 || (b[i].have & 96) != 96) { global_mismatch++; continue; }   // a dump without the fields does not pass the strict form
            global_compared++;
            if (a[i].ga != b[i].ga || a[i].gb != b[i].gb) global_mismatch++;
        }
    const bool same = same_sel && global_mismatch == 0;
    std::printf("{\"identical_selection\": %s, \"identical_dump\": %s, \"global_index_compared\": %zu, \"global_index_mismatches\": %zu, "
                "\"n_reference\": %zu, \"n_candidate\": %zu, \"matched_prefix\": %zu, \"max_abs_score_diff\": %.17g",
                same_sel ? "true" : "false", with_global ? (same ? "true" : "false") : "null", global_compared, global_mismatch, a.size(), b.size(), first, max_d);
    if (!same_sel) {
        std::printf(", \"first_divergence\": {\"index\": %zu", first);
        auto one = [](const char *name, const std::vector<LoopRec> &v, size_t i) {
            if (i < v.size())
                std::printf(", \"%s\": {\"time_sec_a\": %llu, \"time_nsec_a\": %llu, \"time_sec_b\": %llu, \"time_nsec_b\": %llu, \"score\": %.17g}", name,
                            (unsigned long long)v[i].sa, (unsigned long long)v[i].na, (unsigned long long)v[i].sb, (unsigned long long)v[i].nb, v[i].score);
            else std::printf(", \"%s\": null", name);
        };
        one("reference", a, first);
        one("candidate", b, first);
        std::printf("}");
    }
    std::printf("}\n");
    return same ? 0 : 1;
}

// --gate <in.txt> : no GPU.  One candidate per line: sec1 nsec1 sec2 nsec2 idx1 idx2 pf_matches g1 g2 g3 then 3 x 16 doubles
// (column-major op1, op2, icp).  Prints one JSON object per line with the gate decision and the LoopEdge fields.
static int gate(const char *in)
{
    FILE *f = std::fopen(in, "r");
    if (!f) { std::perror(in); return 2; }
    for (;;) {
        cerebro_hip::ProcessedLoopCandidate pc;
        unsigned s1, n1, s2, n2;
        float g[3];
        if (std::fscanf(f, "%u %u %u %u %d %d %d %f %f %f", &s1, &n1, &s2, &n2, &pc.idx_from_datamanager_1, &pc.idx_from_datamanager_2,
                        &pc.pf_matches, &g[0], &g[1], &g[2]) != 10) break;
        pc.t_node_1.sec = s1; pc.t_node_1.nsec = n1; pc.t_node_2.sec = s2; pc.t_node_2.nsec = n2;
        for (int k = 0; k < 3; k++) {
            std::array<double, 16> T;
            for (int e = 0; e < 16; e++)
                if (std::fscanf(f, "%lf", &T[e]) != 1) { std::fprintf(stderr, "bad matrix\n"); return 2; }
            pc.opX_b_T_a.push_back(T);
            pc.opX_goodness.push_back(g[k]);
        }
        cerebro_hip::LoopEdgePOD e;
        const bool ok = pc.makeLoopEdgeMsgWithConsistencyCheck(e);
        if (!ok) { std::printf("{\"ok\": false}\n"); continue; }
        std::printf("{\"ok\": true, \"sec0\": %u, \"nsec0\": %u, \"sec1\": %u, \"nsec1\": %u, \"position\": [%.17g, %.17g, %.17g], "
                    "\"orientation_xyzw\": [%.17g, %.17g, %.17g, %.17g], \"weight\": %.9g, \"description\": \"%s\"}\n",
                    e.timestamp0.sec, e.timestamp0.nsec, e.timestamp1.sec, e.timestamp1.nsec, e.position[0], e.position[1], e.position[2],
                    e.orientation_xyzw[0], e.orientation_xyzw[1], e.orientation_xyzw[2], e.orientation_xyzw[3], (double)e.weight, e.description.c_str());
    }
    std::fclose(f);
    return 0;
}

// --threeway <in.bin> [device] : GPU.  in.bin: u32 N, u32 pf_matches, u64 seed, then N x {3,2,3,2,3,3} f64 arrays
// (world_point_uv, feature_position_uv_d, world_point_uv_d, feature_position_uv, uv_X, uvd_Y).  Runs the three estimators
// and the consistency gate exactly as the loop-candidate consumer does and prints the result as JSON.
static int threeway(const char *in, int device)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) { std::perror(in); return 2; }
    uint32_t N = 0, pf = 0;
    uint64_t seed = 0;
    if (std::fread(&N, 4, 1, f) != 1 || std::fread(&pf, 4, 1, f) != 1 || std::fread(&seed, 8, 1, f) != 1) return 2;
    cerebro_hip::PosePairInput inp;
    inp.world_point_uv.resize(N); inp.feature_position_uv_d.resize(N); inp.world_point_uv_d.resize(N);
    inp.feature_position_uv.resize(N); inp.uv_X.resize(N); inp.uvd_Y.resize(N);
    bool ok = std::fread(inp.world_point_uv.data(), 24, N, f) == N && std::fread(inp.feature_position_uv_d.data(), 16, N, f) == N &&
              std::fread(inp.world_point_uv_d.data(), 24, N, f) == N && std::fread(inp.feature_position_uv.data(), 16, N, f) == N &&
              std::fread(inp.uv_X.data(), 24, N, f) == N && std::fread(inp.uvd_Y.data(), 24, N, f) == N;
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "truncated input\n"); return 2; }
    chip_ctx *ctx = nullptr;
    const int st = chip_create(&ctx, 64, 0, device, 0, 1);
    if (st != CHIP_OK) { std::fprintf(stderr, "chip_create failed: %s\n", chip_strerror(st)); return 3; }
    cerebro_hip::ProcessedLoopCandidate pc;
    pc.t_node_1.sec = 1403636700; pc.t_node_2.sec = 1403636600;
    pc.idx_from_datamanager_1 = 2100; pc.idx_from_datamanager_2 = 100;
    pc.pf_matches = (int)pf;
    const bool have = cerebro_hip::compute_three_way_pose(ctx, inp, pc, seed);
    std::printf("{\"have_poses\": %s", have ? "true" : "false");
    if (have) {
        std::printf(", \"goodness\": [%.9g, %.9g, %.9g], \"poses\": [", (double)pc.opX_goodness[0], (double)pc.opX_goodness[1], (double)pc.opX_goodness[2]);
        for (int k = 0; k < 3; k++) {
            std::printf("%s[", k ? ", " : "");
            for (int e = 0; e < 16; e++) std::printf("%s%.17g", e ? ", " : "", pc.opX_b_T_a[k][e]);
            std::printf("]");
        }
        cerebro_hip::LoopEdgePOD e;
        const bool pub = pc.makeLoopEdgeMsgWithConsistencyCheck(e);
        std::printf("], \"publish\": %s", pub ? "true" : "false");
        if (pub)
            std::printf(", \"position\": [%.17g, %.17g, %.17g], \"orientation_xyzw\": [%.17g, %.17g, %.17g, %.17g], \"weight\": %.9g, \"description\": \"%s\"",
                        e.position[0], e.position[1], e.position[2], e.orientation_xyzw[0], e.orientation_xyzw[1], e.orientation_xyzw[2],
                        e.orientation_xyzw[3], (double)e.weight, e.description.c_str());
    }
    std::printf("}\n");
    chip_destroy(ctx);
    return 0;
}

// --policy naive|clique <stream.bin> <out.json> [device] : same stream format, but each tick runs one iteration of
// faiss__naive_loopcandidate_generator / faiss_clique_loopcandidate_generator instead of descrip_N__dot__descrip_0_N.
// rand() of the clique policy is replaced by the ANSI C example generator seeded with 1 so replays are reproducible.
static unsigned long g_rand_next = 1;
static int replay_rand() { g_rand_next = g_rand_next * 1103515245ul + 12345ul; return (int)((g_rand_next / 65536ul) % 32768ul); }

int main(int argc, char **argv)
{
    if (argc >= 3 && std::strcmp(argv[1], "--devices") == 0) {
        for (const char *p = argv[2]; *p;) {
            g_devices.push_back(std::atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        argv += 2;
        argc -= 2;
    }
    int policy = 0;
    if (argc >= 5 && std::strcmp(argv[1], "--policy") == 0) {
        policy = std::strcmp(argv[2], "naive") == 0 ? 1 : std::strcmp(argv[2], "clique") == 0 ? 2 : -1;
        if (policy < 0) { std::fprintf(stderr, "unknown policy %s\n", argv[2]); return 2; }
        argv += 2;
        argc -= 2;
    }
    if (argc >= 4 && std::strcmp(argv[1], "--compare") == 0) return compare_dumps(argv[2], argv[3], true);
    if (argc >= 4 && std::strcmp(argv[1], "--compare-selection") == 0) return compare_dumps(argv[2], argv[3], false);
    if (argc >= 3 && std::strcmp(argv[1], "--gate") == 0) return gate(argv[2]);
    if (argc >= 3 && std::strcmp(argv[1], "--threeway") == 0) return threeway(argv[2], argc > 3 ? std::atoi(argv[3]) : 0);
    if (argc >= 4 && std::strcmp(argv[1], "--parse-only") == 0) return parse_only(argv[2], argv[3]);
    if (argc >= 4 && std::strcmp(argv[1], "--state") == 0) return from_state(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 0);
    if (argc < 3) { std::fprintf(stderr, "usage: %s [--policy naive|clique] <stream.bin> <out.json> [device] | --state <state.json> <out.json> [device] | --parse-only <state.json> <out.bin>\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    char magic[4];
    uint32_t ver = 0, D = 0;
    uint64_t N = 0, T = 0;
    if (std::fread(magic, 1, 4, f) != 4 || std::memcmp(magic, "CRBR", 4) != 0 || std::fread(&ver, 4, 1, f) != 1 || ver != 1 ||
        std::fread(&D, 4, 1, f) != 1 || std::fread(&N, 8, 1, f) != 1 || std::fread(&T, 8, 1, f) != 1) {
        std::fprintf(stderr, "bad header\n");
        return 2;
    }
    std::vector<cerebro_hip::Time> stamps(N);
    std::vector<double> desc((size_t)N * D);
    std::vector<int64_t> ticks(T);
    if (std::fread(stamps.data(), sizeof(cerebro_hip::Time), N, f) != N || std::fread(desc.data(), sizeof(double), (size_t)N * D, f) != (size_t)N * D ||
        std::fread(ticks.data(), sizeof(int64_t), T, f) != T) {
        std::fprintf(stderr, "truncated stream\n");
        return 2;
    }
    std::vector<cerebro_hip::Time> frames;
    std::vector<int64_t> row_to_frame;
    for (char tag[4]; std::fread(tag, 1, 4, f) == 4;) {      // optional trailers
        uint64_t cnt = 0;
        if (std::fread(&cnt, 8, 1, f) != 1 || cnt > (1ull << 32)) { std::fprintf(stderr, "bad trailer\n"); return 2; }
        if (std::memcmp(tag, "FRMS", 4) == 0) {
            frames.resize(cnt);
            if (std::fread(frames.data(), sizeof(cerebro_hip::Time), cnt, f) != cnt) { std::fprintf(stderr, "truncated FRMS trailer\n"); return 2; }
        } else if (std::memcmp(tag, "FIDX", 4) == 0) {
            row_to_frame.resize(cnt);
            if (std::fread(row_to_frame.data(), sizeof(int64_t), cnt, f) != cnt) { std::fprintf(stderr, "truncated FIDX trailer\n"); return 2; }
        } else { std::fprintf(stderr, "unknown trailer\n"); return 2; }
    }
    std::fclose(f);
    std::unique_ptr<cerebro_hip::Cerebro> cer_p(make_cerebro((int)D, argc > 3 ? std::atoi(argv[3]) : 0, (int64_t)N));
    cerebro_hip::Cerebro &cer = *cer_p;
    if (!cer.ok()) { std::fprintf(stderr, "chip_create failed: %s\n", chip_strerror(cer.last_status())); return 3; }
    cer.rand_source = replay_rand;
    for (const cerebro_hip::Time &t : frames) cer.data_map_insert(t);
    if (!row_to_frame.empty()) cer.set_frame_index_map(row_to_frame);
    int64_t appended = 0;
    for (uint64_t t = 0; t < T; t++) {
        const int64_t l = ticks[t];
        for (; appended < l && appended < (int64_t)N; appended++)
            if (!cer.descriptor_available(stamps[appended], &desc[(size_t)appended * D], (int)D)) {
                std::fprintf(stderr, "append failed at row %lld: %s\n", (long long)appended, chip_strerror(cer.last_status()));
                return 4;
            }
        if (policy == 1) cer.faiss__naive_loopcandidate_generator_once();
        else if (policy == 2) cer.faiss_clique_loopcandidate_generator_once();
        else cer.descrip_N__dot__descrip_0_N_once();
        if (cer.last_status() != CHIP_OK) { std::fprintf(stderr, "tick failed: %s\n", chip_strerror(cer.last_status())); return 5; }
    }
    FILE *o = std::fopen(argv[2], "w");
    if (!o) { std::perror(argv[2]); return 2; }
    const std::string js = cer.foundLoops_as_JSON();
    std::fwrite(js.data(), 1, js.size(), o);
    std::fclose(o);
    std::fprintf(stderr, "cerebro_replay: %llu descriptors, %llu ticks, %d loop candidates\n", (unsigned long long)N, (unsigned long long)T, cer.foundLoops_count());
    return 0;
}

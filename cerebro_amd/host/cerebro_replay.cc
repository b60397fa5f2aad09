// cerebro_replay -- offline replay harness of the loop-candidate producer (BASELINE configs 1 and 5: "harness ready,
// data absent": feed it the descriptors of an EuRoC run and a tick schedule and it reproduces foundLoops).
//
//   cerebro_replay <stream.bin> <out.json> [device]
// stream.bin (little endian): "CRBR" u32 version=1, u32 D, u64 N, u64 n_ticks, then N x {u32 sec, u32 nsec},
// N x D float64 descriptors (the .srv wire type), n_ticks x i64 l (value of wholeImageComputedList_size() at each
// iteration of the dot-product thread; rows < l are appended before the tick).
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

static int from_state(const char *in, const char *outp, int device)
{
    cerebro_hip::StateDescriptors sd;
    if (!cerebro_hip::load_state_json(in, sd)) { std::fprintf(stderr, "parse error: %s\n", sd.error.c_str()); return 6; }
    cerebro_hip::Cerebro cer(sd.D, device, (int64_t)sd.stampNSec.size());
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

int main(int argc, char **argv)
{
    if (argc >= 4 && std::strcmp(argv[1], "--parse-only") == 0) return parse_only(argv[2], argv[3]);
    if (argc >= 4 && std::strcmp(argv[1], "--state") == 0) return from_state(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 0);
    if (argc < 3) { std::fprintf(stderr, "usage: %s <stream.bin> <out.json> [device] | --state <state.json> <out.json> [device] | --parse-only <state.json> <out.bin>\n", argv[0]); return 2; }
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
    std::fclose(f);
    cerebro_hip::Cerebro cer((int)D, argc > 3 ? std::atoi(argv[3]) : 0, (int64_t)N);
    if (!cer.ok()) { std::fprintf(stderr, "chip_create failed: %s\n", chip_strerror(cer.last_status())); return 3; }
    int64_t appended = 0;
    for (uint64_t t = 0; t < T; t++) {
        const int64_t l = ticks[t];
        for (; appended < l && appended < (int64_t)N; appended++)
            if (!cer.descriptor_available(stamps[appended], &desc[(size_t)appended * D], (int)D)) {
                std::fprintf(stderr, "append failed at row %lld: %s\n", (long long)appended, chip_strerror(cer.last_status()));
                return 4;
            }
        cer.descrip_N__dot__descrip_0_N_once();
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

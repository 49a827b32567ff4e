// Host -> host cost of Parse() through the C ABI, split into its parts (GPU box):
//   g++ -O2 -std=c++17 tools/h2h_bench.cpp -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ \
//       -L simdjson-go_amd -lsjhip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/simdjson-go_amd -o /tmp/h2h_bench
//   /tmp/h2h_bench tests/fixtures/twitter.json [ndjson]
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "sjhip.h"

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename F>
static double best_of(int reps, int inner, F f) {
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        const double t0 = now_us();
        for (int k = 0; k < inner; k++) f();
        const double dt = (now_us() - t0) / inner;
        if (dt < best) best = dt;
    }
    return best;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> doc;
    {
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) doc.insert(doc.end(), buf, buf + n);
        fclose(f);
    }
    const uint32_t flags = SJHIP_FLAG_COPY_STRINGS | (argc > 2 ? SJHIP_FLAG_NDJSON : 0u);
    sjhip_ctx *ctx = sjhip_ctx_create(0);
    if (!ctx) return 3;
    size_t tl = 0, sl = 0, mo = 0, ml = 0;
    int rc = sjhip_parse(ctx, doc.data(), doc.size(), flags, &tl, &sl, &mo, &ml);
    if (rc) {
        printf("parse rc=%d %s\n", rc, sjhip_last_error(ctx));
        return 4;
    }
    std::vector<uint64_t> tape(tl);
    std::vector<uint8_t> strings(sl + 1);
    const double t_parse = best_of(5, 40, [&] { sjhip_parse(ctx, doc.data(), doc.size(), flags, &tl, &sl, &mo, &ml); });
    const double t_both = best_of(5, 40, [&] {
        sjhip_parse(ctx, doc.data(), doc.size(), flags, &tl, &sl, &mo, &ml);
        sjhip_fetch(ctx, tape.data(), strings.data());
    });
    const uint64_t *vt = nullptr;
    const uint8_t *vs = nullptr;
    uint64_t sink = 0;
    const double t_view = best_of(5, 40, [&] {
        sjhip_parse(ctx, doc.data(), doc.size(), flags, &tl, &sl, &mo, &ml);
        sjhip_fetch_view(ctx, &vt, &vs);
        sink += vt ? vt[tl - 1] : 0;
    });
    uint8_t *in = sjhip_input_block(ctx, doc.size());
    const double t_pin_view = in ? best_of(5, 40, [&] {
        memcpy(in, doc.data(), doc.size());  // (stands for the caller's read into the block)
        sjhip_parse(ctx, in, doc.size(), flags, &tl, &sl, &mo, &ml);
        sjhip_fetch_view(ctx, &vt, &vs);
    }) : 0.0;
    const double t_pin_view_nc = in ? best_of(5, 40, [&] {
        sjhip_parse(ctx, in, doc.size(), flags, &tl, &sl, &mo, &ml);
        sjhip_fetch_view(ctx, &vt, &vs);
    }) : 0.0;
    printf("%s: %zu B, tape %zu words, strings %zu B\n", argv[1], doc.size(), tl, sl);
    printf("  input block (memcpy in) + parse + view     %8.1f us\n", t_pin_view);
    printf("  input block (already filled) + parse + view %7.1f us\n", t_pin_view_nc);
    printf("  sjhip_parse + sjhip_fetch_view (in place)  %8.1f us   (%llu)\n", t_view, (unsigned long long)(sink & 1));
    printf("  sjhip_parse (pageable H2D + kernels)      %8.1f us\n", t_parse);
    printf("  sjhip_parse + sjhip_fetch (pageable D2H)  %8.1f us\n", t_both);
    // the raw copies of the same sizes
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *d = nullptr, *pin = nullptr;
    const size_t big = doc.size() + tl * 8 + sl + 4096;
    hipMalloc(&d, big);
    hipHostMalloc(&pin, big, hipHostMallocDefault);
    auto cp = [&](void *dst, const void *src, size_t n, hipMemcpyKind k) {
        hipMemcpyAsync(dst, src, n, k, s);
        hipStreamSynchronize(s);
    };
    printf("  H2D message pageable                       %8.1f us\n", best_of(5, 40, [&] { cp(d, doc.data(), doc.size(), hipMemcpyHostToDevice); }));
    printf("  H2D message pinned                         %8.1f us\n", best_of(5, 40, [&] { cp(d, pin, doc.size(), hipMemcpyHostToDevice); }));
    printf("  memcpy message -> pinned + H2D             %8.1f us\n", best_of(5, 40, [&] {
               memcpy(pin, doc.data(), doc.size());
               cp(d, pin, doc.size(), hipMemcpyHostToDevice);
           }));
    printf("  D2H tape+strings pageable (2 copies)       %8.1f us\n", best_of(5, 40, [&] {
               hipMemcpyAsync(tape.data(), d, tl * 8, hipMemcpyDeviceToHost, s);
               hipMemcpyAsync(strings.data(), (char *)d + tl * 8, sl, hipMemcpyDeviceToHost, s);
               hipStreamSynchronize(s);
           }));
    printf("  D2H tape+strings pinned (2 copies)         %8.1f us\n", best_of(5, 40, [&] {
               hipMemcpyAsync(pin, d, tl * 8, hipMemcpyDeviceToHost, s);
               hipMemcpyAsync((char *)pin + tl * 8, (char *)d + tl * 8, sl, hipMemcpyDeviceToHost, s);
               hipStreamSynchronize(s);
           }));
    printf("  D2H pinned (1 copy) + 2 memcpy to pageable %8.1f us\n", best_of(5, 40, [&] {
               hipMemcpyAsync(pin, d, tl * 8 + sl, hipMemcpyDeviceToHost, s);
               hipStreamSynchronize(s);
               memcpy(tape.data(), pin, tl * 8);
               memcpy(strings.data(), (char *)pin + tl * 8, sl);
           }));
    printf("  memcpy tape+strings pinned -> pageable     %8.1f us\n", best_of(5, 40, [&] {
               memcpy(tape.data(), pin, tl * 8);
               memcpy(strings.data(), (char *)pin + tl * 8, sl);
           }));
    printf("  empty stream sync                          %8.1f us\n", best_of(5, 40, [&] { hipStreamSynchronize(s); }));
    sjhip_ctx_destroy(ctx);
    return 0;
}

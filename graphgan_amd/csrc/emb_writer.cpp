// emb_writer.cpp -- the reference's embedding file format, natively.
// Replaces the text formatting of GraphGAN.write_embeddings_to_file (reference
// src/GraphGAN/graph_gan.py:293-306): header "N\td\n", then per node "id\tv0\tv1...\n" where every
// value is the fp32 embedding widened to fp64 and printed with Python's str(float) (= repr: the
// shortest string that round-trips, in CPython's notation rules).  Byte-identical output, formatted
// by host threads (std::to_chars gives the shortest digits; the notation rules are restated from
// CPython's format_float_short: exponent form iff decpt <= -4 or decpt > 16, exponent with sign
// and at least two digits, fixed form always with a fractional part).
#include <charconv>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gg_internal.h"

namespace gg {

// Python repr(float(v)); returns the end of the written text.  out needs >= 32 bytes.
char *py_float_repr(double v, char *out) {
    if (std::isnan(v)) { memcpy(out, "nan", 3); return out + 3; }
    if (std::isinf(v)) {
        if (v < 0) *out++ = '-';
        memcpy(out, "inf", 3);
        return out + 3;
    }
    if (v == 0.0) {
        if (std::signbit(v)) *out++ = '-';
        memcpy(out, "0.0", 3);
        return out + 3;
    }
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX, shortest
    const char *p = buf, *end = r.ptr;
    if (*p == '-') { *out++ = '-'; ++p; }
    char digits[24];
    int nd = 0;
    while (p < end && *p != 'e') {
        if (*p != '.') digits[nd++] = *p;
        ++p;
    }
    ++p;  // 'e'
    int esign = 1;
    if (*p == '-') { esign = -1; ++p; } else if (*p == '+') ++p;
    int e = 0;
    while (p < end) e = e * 10 + (*p++ - '0');
    e *= esign;
    const int decpt = e + 1;  // value = 0.d1d2... * 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *out++ = digits[0];
        if (nd > 1) {
            *out++ = '.';
            memcpy(out, digits + 1, nd - 1);
            out += nd - 1;
        }
        *out++ = 'e';
        *out++ = e < 0 ? '-' : '+';
        int ae = e < 0 ? -e : e;
        if (ae >= 100) { *out++ = (char)('0' + ae / 100); ae %= 100; }
        *out++ = (char)('0' + ae / 10);
        *out++ = (char)('0' + ae % 10);
    } else if (decpt <= 0) {
        *out++ = '0';
        *out++ = '.';
        for (int i = 0; i < -decpt; ++i) *out++ = '0';
        memcpy(out, digits, nd);
        out += nd;
    } else if (decpt < nd) {
        memcpy(out, digits, decpt);
        out += decpt;
        *out++ = '.';
        memcpy(out, digits + decpt, nd - decpt);
        out += nd - decpt;
    } else {
        memcpy(out, digits, nd);
        out += nd;
        for (int i = 0; i < decpt - nd; ++i) *out++ = '0';
        *out++ = '.';
        *out++ = '0';
    }
    return out;
}

static void format_rows(const float *emb, int64_t r0, int64_t r1, int d, std::string &dst) {
    dst.clear();
    dst.reserve((size_t)(r1 - r0) * (12 + 24 * (size_t)d));
    char tmp[40];
    for (int64_t r = r0; r < r1; ++r) {
        dst += std::to_string(r);
        const float *row = emb + r * d;
        for (int j = 0; j < d; ++j) {
            dst += '\t';
            char *e = py_float_repr((double)row[j], tmp);
            dst.append(tmp, e - tmp);
        }
        dst += '\n';
    }
}

int write_embedding_file(gg_ctx *ctx, const float *emb, int64_t n, int d, const char *path, int n_threads) {
    FILE *f = fopen(path, "w");
    if (!f) return fail(ctx, GG_EIO, "cannot open %s for writing", path);
    fprintf(f, "%lld\t%d\n", (long long)n, d);
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    if (n_threads > 64) n_threads = 64;
    const int64_t chunk = 2048;
    int rc = GG_OK;
    for (int64_t base = 0; base < n && rc == GG_OK; base += chunk * n_threads) {
        const int nt = (int)std::min<int64_t>(n_threads, (n - base + chunk - 1) / chunk);
        std::vector<std::string> parts(nt);
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t)
            th.emplace_back([&, t]() { format_rows(emb, base + t * chunk, std::min<int64_t>(n, base + (t + 1) * chunk), d, parts[t]); });
        format_rows(emb, base, std::min<int64_t>(n, base + chunk), d, parts[0]);
        for (auto &x : th) x.join();
        for (int t = 0; t < nt; ++t)
            if (fwrite(parts[t].data(), 1, parts[t].size(), f) != parts[t].size()) rc = fail(ctx, GG_EIO, "short write to %s", path);
    }
    if (fclose(f) != 0 && rc == GG_OK) rc = fail(ctx, GG_EIO, "close of %s failed", path);
    return rc;
}

}  // namespace gg

using namespace gg;

extern "C" int gg_host_write_embeddings(const float *emb, int64_t n_node, int32_t n_emb, const char *path, int32_t n_threads) {
    if (!emb || !path || n_node < 0 || n_emb <= 0) return fail(nullptr, GG_EINVAL, "gg_host_write_embeddings: bad argument");
    return write_embedding_file(nullptr, emb, n_node, n_emb, path, n_threads);
}

extern "C" int gg_write_embeddings(gg_ctx *ctx, int32_t which, const char *path, int32_t n_threads) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, (which == 0 || which == 1) && path, GG_EINVAL, "gg_write_embeddings: bad argument");
    std::vector<float> host((size_t)ctx->n_node * ctx->n_emb);
    int rc = gg_get_embeddings(ctx, which, host.data());
    if (rc != GG_OK) return rc;
    return write_embedding_file(ctx, host.data(), ctx->n_node, ctx->n_emb, path, n_threads);
}

// Binary side-car of the .emb text: at N = 10^7, d = 256 the reference's text is ~50 GB per
// model and write; the side-car holds the same fp32 numbers in 10 GB.  Layout: {magic "GGEB", version 1, n_emb i32,
// n_node i64} then n_node * n_emb fp32, row-major, little endian (numpy: np.fromfile(f, "<f4", offset=20)).
extern "C" int gg_write_embeddings_bin(gg_ctx *ctx, int32_t which, const char *path) {
    if (!ctx) return fail(nullptr, GG_EINVAL, "ctx is NULL");
    GG_CHECK(ctx, (which == 0 || which == 1) && path, GG_EINVAL, "gg_write_embeddings_bin: bad argument");
    std::vector<float> host((size_t)ctx->n_node * ctx->n_emb);
    int rc = gg_get_embeddings(ctx, which, host.data());
    if (rc != GG_OK) return rc;
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(ctx, GG_EIO, "gg_write_embeddings_bin: cannot open %s", tmp.c_str());
    struct { char magic[4]; int32_t version, n_emb; int64_t n_node; } __attribute__((packed)) h = {{'G', 'G', 'E', 'B'}, 1, ctx->n_emb, ctx->n_node};
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(host.data(), sizeof(float), host.size(), f) == host.size() && fflush(f) == 0;
    fclose(f);
    if (!ok || rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        return fail(ctx, GG_EIO, "gg_write_embeddings_bin: cannot write %s", path);
    }
    return GG_OK;
}

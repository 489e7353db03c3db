// ingest.cpp -- edge-list ingest.  Replaces utils.read_edges /
// read_edges_from_file (reference src/utils.py:12-54) for large files: one pass over the text,
// adjacency in CSR with the reference's list order -- for train edge k = (a, b): b is appended to
// graph[a], then a to graph[b] (a self-loop lists a twice); nodes that occur only in the test file
// get empty lists; n_node = number of distinct ids (ids must be 0..n_node-1, README.md:32-39).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gg_internal.h"

namespace {

// whitespace-separated integers of a file, two per line (like line.split() + int()); blank lines skipped
int read_pairs(const char *path, std::vector<int64_t> &out, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return GG_EIO; }
    std::vector<char> buf(1 << 22);
    int64_t cur = 0, first = 0;
    bool in_num = false, neg = false;
    int on_line = 0;
    int64_t line = 1;
    size_t got;
    auto end_token = [&]() {
        if (!in_num) return;
        const int64_t v = neg ? -cur : cur;
        if (on_line == 0) first = v;
        else if (on_line == 1) { out.push_back(first); out.push_back(v); }
        ++on_line;
        in_num = false; neg = false; cur = 0;
    };
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (c >= '0' && c <= '9') { cur = cur * 10 + (c - '0'); in_num = true; }
            else if (c == '-' && !in_num) { neg = true; in_num = true; }
            else if (c == '\n') {
                end_token();
                if (on_line == 1) { fclose(f); err = std::string(path) + ": line " + std::to_string(line) + " has one id"; return GG_EINVAL; }
                on_line = 0;
                ++line;
            } else if (c == ' ' || c == '\t' || c == '\r') end_token();
            else { fclose(f); err = std::string(path) + ": line " + std::to_string(line) + ": unexpected character"; return GG_EINVAL; }
        }
    }
    end_token();
    fclose(f);
    if (on_line == 1) { err = std::string(path) + ": last line has one id"; return GG_EINVAL; }
    return GG_OK;
}

}  // namespace

extern "C" {

int gg_host_read_edges(const char *train_path, const char *test_path, gg_graph *out) {
    if (!train_path || !out) return gg::fail(nullptr, GG_EINVAL, "gg_host_read_edges: bad argument");
    memset(out, 0, sizeof(*out));
    std::vector<int64_t> train, test;
    std::string err;
    int rc = read_pairs(train_path, train, err);
    if (rc == GG_OK && test_path && test_path[0]) rc = read_pairs(test_path, test, err);
    if (rc != GG_OK) return gg::fail(nullptr, rc, "gg_host_read_edges: %s", err.c_str());
    int64_t max_id = -1;
    for (int64_t v : train) { if (v < 0) return gg::fail(nullptr, GG_EINVAL, "gg_host_read_edges: negative id"); if (v > max_id) max_id = v; }
    for (int64_t v : test) { if (v < 0) return gg::fail(nullptr, GG_EINVAL, "gg_host_read_edges: negative id"); if (v > max_id) max_id = v; }
    if (max_id >= (1ll << 31) - 1) return gg::fail(nullptr, GG_EINVAL, "gg_host_read_edges: id %lld does not fit int32", (long long)max_id);
    const int64_t n = max_id + 1;
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int64_t v : train) seen[v] = 1;
    for (int64_t v : test) seen[v] = 1;
    int64_t distinct = 0;
    for (int64_t v = 0; v < n; ++v) distinct += seen[v];
    // the reference takes n_node = len(set of ids) and then indexes 0..n_node-1: ids must be dense
    if (distinct != n) return gg::fail(nullptr, GG_EINVAL, "gg_host_read_edges: ids are not 0..N-1 (%lld distinct ids, largest %lld)", (long long)distinct, (long long)max_id);
    int64_t *rowptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    const int64_t nnz = (int64_t)train.size();
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (!rowptr || !col) { free(rowptr); free(col); return gg::fail(nullptr, GG_ENOMEM, "gg_host_read_edges: out of memory"); }
    for (size_t i = 0; i < train.size(); i += 2) { rowptr[train[i] + 1] += 1; rowptr[train[i + 1] + 1] += 1; }
    for (int64_t v = 0; v < n; ++v) rowptr[v + 1] += rowptr[v];
    std::vector<int64_t> fill(rowptr, rowptr + n);
    for (size_t i = 0; i < train.size(); i += 2) {
        const int64_t a = train[i], b = train[i + 1];
        col[fill[a]++] = (int32_t)b;  // graph[a].append(b)
        col[fill[b]++] = (int32_t)a;  // graph[b].append(a)
    }
    out->n_node = (int32_t)n;
    out->nnz = nnz;
    out->n_train_edges = (int64_t)train.size() / 2;
    out->n_test_edges = (int64_t)test.size() / 2;
    out->rowptr = rowptr;
    out->col = col;
    return GG_OK;
}

void gg_host_free_graph(gg_graph *g) {
    if (!g) return;
    free(g->rowptr);
    free(g->col);
    memset(g, 0, sizeof(*g));
}

}  // extern "C"

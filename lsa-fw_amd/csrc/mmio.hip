// Host-side MatrixMarket coordinate reader (the on-disk stage boundary of the path: A.mtx / M.mtx written by the
// reference's assembly stage, FEM/utils.py:616-636, and read back by .examples/eigenvalues.py:74-77 through scipy's
// mmread plus a per-entry PETSc setValue loop, FEM/utils.py:143-147,208-215).  Parses the whole file in one pass into
// CSR: general / symmetric / hermitian / skew-symmetric storage, real / integer / complex / pattern fields, explicit
// zeros kept, duplicate coordinates summed, columns sorted.  No device code; compiled into liblsa_hip.so so that the
// Python front end needs one library.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdlib>

#include "lsa_internal.h"

struct lsa_mm {
    int32_t nrows = 0, ncols = 0;
    int is_complex = 0;
    std::vector<int32_t> rp, ci;
    std::vector<double> val;  // interleaved (re, im) when complex
    std::string err;
};

namespace {

inline const char* skip_ws(const char* p, const char* e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    return p;
}

inline bool parse_long(const char*& p, const char* e, long& out) {
    p = skip_ws(p, e);
    if (p >= e) return false;
    char* q = nullptr;
    errno = 0;
    out = strtol(p, &q, 10);
    if (q == p || errno) return false;
    p = q;
    return true;
}

inline bool parse_double(const char*& p, const char* e, double& out) {
    p = skip_ws(p, e);
    if (p >= e) return false;
    char* q = nullptr;
    out = strtod(p, &q);
    if (q == p) return false;
    p = q;
    return true;
}

std::string lower(std::string s) {
    for (char& c : s) c = (char)tolower((unsigned char)c);
    return s;
}

}  // namespace

extern "C" {

void lsa_mm_close(lsa_mm* h) { delete h; }

const char* lsa_mm_error(const lsa_mm* h) { return h ? h->err.c_str() : "null handle"; }

// Parse `path`.  On success *out holds the CSR; shape / nnz / scalar kind are returned through the pointers.
int lsa_mm_open(const char* path, lsa_mm** out, int32_t* nrows, int32_t* ncols, int64_t* nnz, int* is_complex) {
    if (!path || !out) return LSA_ERR_ARG;
    lsa_mm* h = new lsa_mm();
    *out = h;
    FILE* f = fopen(path, "rb");
    if (!f) {
        h->err = std::string("cannot open ") + path;
        return LSA_ERR_ARG;
    }
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)size + 1);
    const size_t got = fread(buf.data(), 1, (size_t)size, f);
    fclose(f);
    buf[got] = '\0';
    const char* p = buf.data();
    const char* e = p + got;
    // banner
    const char* eol = (const char*)memchr(p, '\n', (size_t)(e - p));
    if (!eol) eol = e;
    std::string banner = lower(std::string(p, eol));
    char b0[64], b1[64], fmt[64], field[64], sym[64];
    if (sscanf(banner.c_str(), "%63s %63s %63s %63s %63s", b0, b1, fmt, field, sym) != 5 || std::string(b0) != "%%matrixmarket" ||
        std::string(b1) != "matrix") {
        h->err = "not a MatrixMarket matrix file";
        return LSA_ERR_ARG;
    }
    if (std::string(fmt) != "coordinate") {
        h->err = "only coordinate format is supported";
        return LSA_ERR_ARG;
    }
    const std::string fld = field, sy = sym;
    const bool cplx = fld == "complex", pattern = fld == "pattern";
    if (!cplx && !pattern && fld != "real" && fld != "integer" && fld != "double") {
        h->err = "unsupported field " + fld;
        return LSA_ERR_ARG;
    }
    const bool general = sy == "general", symm = sy == "symmetric", herm = sy == "hermitian", skew = sy == "skew-symmetric";
    if (!general && !symm && !herm && !skew) {
        h->err = "unsupported symmetry " + sy;
        return LSA_ERR_ARG;
    }
    p = eol < e ? eol + 1 : e;
    while (p < e) {  // comment / blank lines
        const char* q = skip_ws(p, e);
        if (q < e && *q == '%') {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
            p = nl ? nl + 1 : e;
        } else {
            p = q;
            break;
        }
    }
    long nr = 0, nc = 0, nz = 0;
    if (!parse_long(p, e, nr) || !parse_long(p, e, nc) || !parse_long(p, e, nz) || nr < 0 || nc < 0 || nz < 0 || nr >= INT32_MAX ||
        nc >= INT32_MAX) {
        h->err = "bad size line";
        return LSA_ERR_ARG;
    }
    struct Entry {
        int32_t r, c;
        double re, im;
    };
    // an entry takes at least four bytes of text ("1 1\n"): a size line that promises more than the file can hold is
    // rejected before anything is reserved for it (no bad_alloc across the C boundary)
    if ((uint64_t)nz > (uint64_t)(e - p) / 4 + 1) {
        h->err = "the size line announces more entries than the file holds";
        return LSA_ERR_ARG;
    }
    std::vector<Entry> ent;
    ent.reserve((size_t)nz * (general ? 1 : 2));
    for (long k = 0; k < nz; ++k) {
        long r = 0, c = 0;
        double re = 1.0, im = 0.0;
        if (!parse_long(p, e, r) || !parse_long(p, e, c) || (!pattern && !parse_double(p, e, re)) || (cplx && !parse_double(p, e, im)) ||
            r < 1 || r > nr || c < 1 || c > nc) {
            h->err = "bad entry at line " + std::to_string(k + 1) + " of the data section";
            return LSA_ERR_ARG;
        }
        ent.push_back({(int32_t)(r - 1), (int32_t)(c - 1), re, im});
        if (!general && r != c) {
            if (symm) ent.push_back({(int32_t)(c - 1), (int32_t)(r - 1), re, im});
            else if (herm) ent.push_back({(int32_t)(c - 1), (int32_t)(r - 1), re, -im});
            else ent.push_back({(int32_t)(c - 1), (int32_t)(r - 1), -re, -im});
        }
    }
    std::stable_sort(ent.begin(), ent.end(), [](const Entry& a, const Entry& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    h->nrows = (int32_t)nr;
    h->ncols = (int32_t)nc;
    h->is_complex = cplx ? 1 : 0;
    h->rp.assign((size_t)nr + 1, 0);
    h->ci.reserve(ent.size());
    h->val.reserve(ent.size() * (cplx ? 2 : 1));
    for (size_t k = 0; k < ent.size(); ++k) {
        if (k > 0 && ent[k].r == ent[k - 1].r && ent[k].c == ent[k - 1].c) {  // duplicates are summed (as scipy does)
            if (cplx) {
                h->val[h->val.size() - 2] += ent[k].re;
                h->val[h->val.size() - 1] += ent[k].im;
            } else {
                h->val.back() += ent[k].re;
            }
            continue;
        }
        h->ci.push_back(ent[k].c);
        h->val.push_back(ent[k].re);
        if (cplx) h->val.push_back(ent[k].im);
        ++h->rp[(size_t)ent[k].r + 1];
    }
    for (long i = 0; i < nr; ++i) h->rp[(size_t)i + 1] += h->rp[(size_t)i];
    if (h->ci.size() >= (size_t)INT32_MAX) {
        h->err = "more than 2^31 stored entries";
        return LSA_ERR_ARG;
    }
    if (nrows) *nrows = h->nrows;
    if (ncols) *ncols = h->ncols;
    if (nnz) *nnz = (int64_t)h->ci.size();
    if (is_complex) *is_complex = h->is_complex;
    return LSA_OK;
}

// Copy the parsed CSR into caller buffers: rowptr[nrows + 1], col[nnz], val[nnz] (float64, or interleaved complex128).
int lsa_mm_read_csr(const lsa_mm* h, int32_t* rowptr, int32_t* col, void* val) {
    if (!h || !rowptr || (!h->ci.empty() && (!col || !val))) return LSA_ERR_ARG;
    memcpy(rowptr, h->rp.data(), sizeof(int32_t) * h->rp.size());
    if (!h->ci.empty()) {
        memcpy(col, h->ci.data(), sizeof(int32_t) * h->ci.size());
        memcpy(val, h->val.data(), sizeof(double) * h->val.size());
    }
    return LSA_OK;
}

}  // extern "C"

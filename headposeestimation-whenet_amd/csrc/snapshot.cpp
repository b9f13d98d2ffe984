#include "snapshot.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace whenet {

// whenet.py:23-26: img/255 then (img-mean)/std in float64; Keras casts to float32 (whenet.py:27).
void normalise_table(float lut[3][256]) {
    const double mean[3] = {0.485, 0.456, 0.406};
    const double stdv[3] = {0.229, 0.224, 0.225};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile double x = double(v) / 255.0;
            volatile double y = x - mean[c];
            volatile double z = y / stdv[c];
            lut[c][v] = float(z);
        }
}


namespace {

struct Reader {
    const uint8_t* p;
    size_t n, off = 0;
    template <typename T> T get() {
        WHENET_REQUIRE(off + sizeof(T) <= n, WHENET_EFORMAT, "snapshot: truncated table");
        T v;
        std::memcpy(&v, p + off, sizeof(T));
        off += sizeof(T);
        return v;
    }
};

const RawTensor& need(const std::map<std::string, RawTensor>& t, const std::string& name,
                      std::initializer_list<uint32_t> dims) {
    auto it = t.find(name);
    WHENET_REQUIRE(it != t.end(), WHENET_EFORMAT, "snapshot: missing tensor " + name);
    const RawTensor& r = it->second;
    bool ok = r.dims.size() == dims.size();
    size_t i = 0;
    for (uint32_t d : dims) ok = ok && (i < r.dims.size()) && (r.dims[i++] == d);
    if (!ok) {
        std::string got, want;
        for (uint32_t d : r.dims) got += std::to_string(d) + ",";
        for (uint32_t d : dims) want += std::to_string(d) + ",";
        throw Error(WHENET_EFORMAT, "snapshot: " + name + " has shape (" + got + ") expected (" + want + ")");
    }
    return r;
}

// BN folded to (scale, shift):  y = x*scale + shift   (double arithmetic)
struct Folded {
    std::vector<double> scale, shift;
};

Folded fold_bn(const std::map<std::string, RawTensor>& t, const std::string& prefix, uint32_t c) {
    const float* g = need(t, prefix + "/gamma", {c}).data;
    const float* b = need(t, prefix + "/beta", {c}).data;
    const float* m = need(t, prefix + "/mean", {c}).data;
    const float* v = need(t, prefix + "/var", {c}).data;
    Folded f;
    f.scale.resize(c);
    f.shift.resize(c);
    for (uint32_t i = 0; i < c; ++i) {
        WHENET_REQUIRE(double(v[i]) + BN_EPS > 0.0, WHENET_EFORMAT, "snapshot: " + prefix + " variance <= -eps");
        double s = double(g[i]) / std::sqrt(double(v[i]) + BN_EPS);
        f.scale[i] = s;
        f.shift[i] = double(b[i]) - double(m[i]) * s;
    }
    return f;
}

template <typename T>
void pack_pw_t(HostPw& pw, const std::vector<double>& wf /*[K][N]*/) {
    constexpr int V = Vec<T>::V;
    const int K = pw.K, N = pw.N;
    pw.KS = ceil_div(K, 2 * V);
    pw.NTILES = ceil_div(N, 32);
    pw.packed.assign(size_t(pw.KS) * pw.NTILES * 64 * V * sizeof(T), 0);
    T* dst = reinterpret_cast<T*>(pw.packed.data());
    for (int ks = 0; ks < pw.KS; ++ks)
        for (int nt = 0; nt < pw.NTILES; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < V; ++e) {
                    int n = nt * 32 + (lane & 31);
                    int k = ks * 2 * V + (lane >> 5) * V + e;
                    double v = (n < N && k < K) ? wf[size_t(k) * N + n] : 0.0;
                    dst[((size_t(ks) * pw.NTILES + nt) * 64 + lane) * V + e] = T(v);
                }
    pw.dense.resize(size_t(K) * N);
    for (size_t i = 0; i < pw.dense.size(); ++i) pw.dense[i] = float(T(wf[i]));
}

// WHENET_F32S operand images (snapshot.h): the folded weights, scaled by a power of two so that the lo halves of the layer's
// small weights are NORMAL binary16 numbers (|w| ~ 0.05 has lo ~ 1e-5, below 2^-14: its subnormal spacing 2^-24 would cap the
// pair at ~20 bits), split as hi = f16(w'), lo = f16(w' - hi): hi + lo carries 22 bits of w'.
void pack_pw_split(HostPw& pw, const std::vector<double>& wf /*[K][N]*/) {
    const int K = pw.K, N = pw.N;
    double mx = 0.0;
    for (double v : wf) mx = std::max(mx, std::fabs(v));
    int shift = 0;
    if (mx > 0.0) {
        shift = int(std::floor(std::log2(16384.0 / mx)));           // largest |w'| in [8192, 16384): far from 65504
        shift = std::max(-24, std::min(24, shift));
    }
    const double sc = std::ldexp(1.0, shift);
    pw.wsi = float(std::ldexp(1.0, -shift));
    pw.KS_split = ceil_div(K, 16);
    const int NTILES = ceil_div(N, 32);
    const size_t per = size_t(pw.KS_split) * NTILES * 64 * 8;
    pw.packed_split.assign(2 * per * sizeof(half_t), 0);
    half_t* hi = reinterpret_cast<half_t*>(pw.packed_split.data());
    half_t* lo = hi + per;
    for (int ks = 0; ks < pw.KS_split; ++ks)
        for (int nt = 0; nt < NTILES; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int n = nt * 32 + (lane & 31);
                    const int k = ks * 16 + (lane >> 5) * 8 + e;
                    const double v = (n < N && k < K) ? wf[size_t(k) * N + n] * sc : 0.0;
                    const half_t h = half_t(float(v));
                    const half_t l = half_t(float(v - double(float(h))));
                    const size_t i = ((size_t(ks) * NTILES + nt) * 64 + lane) * 8 + e;
                    hi[i] = h;
                    lo[i] = l;
                }
}

// 1x1 conv + BatchNorm as one affine map, in double: wf [K][N] = kernel * bn scale, shift [N]
struct FoldedPw {
    std::vector<double> wf, shift;
};
FoldedPw fold_pw(const std::map<std::string, RawTensor>& t, const std::string& conv, const std::string& bn, uint32_t K,
                 uint32_t N) {
    const float* w = need(t, conv + "/kernel", {1, 1, K, N}).data;   // HWIO, 1x1
    Folded f = fold_bn(t, bn, N);
    FoldedPw o;
    o.wf.resize(size_t(K) * N);
    for (uint32_t k = 0; k < K; ++k)
        for (uint32_t n = 0; n < N; ++n) o.wf[size_t(k) * N + n] = double(w[size_t(k) * N + n]) * f.scale[n];
    o.shift = f.shift;
    return o;
}

HostPw pack_pw(const FoldedPw& f, uint32_t K, uint32_t N, int dtype, bool split = false) {
    HostPw pw;
    pw.K = int(K);
    pw.N = int(N);
    pw.bias.resize(N);
    for (uint32_t n = 0; n < N; ++n) pw.bias[n] = float(f.shift[n]);
    if (dtype == WHENET_F16) pack_pw_t<half_t>(pw, f.wf);
    else pack_pw_t<float>(pw, f.wf);
    if (split) pack_pw_split(pw, f.wf);
    return pw;
}

HostPw make_pw(const std::map<std::string, RawTensor>& t, const std::string& conv, const std::string& bn,
               uint32_t K, uint32_t N, int dtype, bool split = false) {
    return pack_pw(fold_pw(t, conv, bn, K, N), K, N, dtype, split);
}

}  // namespace

std::map<std::string, RawTensor> parse_snapshot(const void* blob, size_t nbytes) {
    WHENET_REQUIRE(blob != nullptr && nbytes >= 24, WHENET_EFORMAT, "snapshot: too small");
    const uint8_t* p = static_cast<const uint8_t*>(blob);
    WHENET_REQUIRE(std::memcmp(p, "WHNPACK1", 8) == 0, WHENET_EFORMAT,
                   "snapshot: not a WHNPACK1 file (convert a Keras .h5 with tools/convert_h5.py)");
    Reader r{p, nbytes, 8};
    uint32_t ver = r.get<uint32_t>();
    uint32_t n = r.get<uint32_t>();
    uint64_t data_off = r.get<uint64_t>();
    WHENET_REQUIRE(ver == 1, WHENET_EFORMAT, "snapshot: unsupported version " + std::to_string(ver));
    WHENET_REQUIRE(data_off <= nbytes && n < 100000, WHENET_EFORMAT, "snapshot: bad header");
    std::map<std::string, RawTensor> out;
    for (uint32_t i = 0; i < n; ++i) {
        uint16_t ln = r.get<uint16_t>();
        WHENET_REQUIRE(r.off + ln <= nbytes, WHENET_EFORMAT, "snapshot: truncated name");
        std::string name(reinterpret_cast<const char*>(p + r.off), ln);
        r.off += ln;
        uint8_t dt = r.get<uint8_t>();
        uint8_t nd = r.get<uint8_t>();
        WHENET_REQUIRE(dt == 0 && nd >= 1 && nd <= 4, WHENET_EFORMAT, "snapshot: " + name + ": unsupported dtype/rank");
        RawTensor t;
        uint64_t count = 1;
        const uint64_t max_count = uint64_t(nbytes) / 4;      // no tensor can hold more elements than the file has
        for (int d = 0; d < nd; ++d) {
            uint32_t x = r.get<uint32_t>();
            t.dims.push_back(x);
            // overflow-safe product: bail out as soon as it exceeds what the blob could hold
            WHENET_REQUIRE(x == 0 || count <= max_count / x, WHENET_EFORMAT,
                           "snapshot: " + name + ": dimensions exceed the file size");
            count *= x;
        }
        uint64_t off = r.get<uint64_t>();
        uint64_t nb = r.get<uint64_t>();
        // every comparison is written so that no intermediate sum can wrap (a crafted offset near
        // 2^64 must not pass): data_off <= nbytes was checked above
        const uint64_t room = uint64_t(nbytes) - data_off;
        WHENET_REQUIRE(nb == count * 4 && off <= room && nb <= room - off && ((data_off + off) % 4) == 0,
                       WHENET_EFORMAT, "snapshot: " + name + ": payload out of bounds");
        t.data = reinterpret_cast<const float*>(p + data_off + off);
        t.count = size_t(count);
        for (size_t j = 0; j < count; ++j)
            WHENET_REQUIRE(std::isfinite(t.data[j]), WHENET_EFORMAT, "snapshot: " + name + ": non-finite value");
        out.emplace(std::move(name), std::move(t));
    }
    return out;
}

HostModel build_host_model(const std::map<std::string, RawTensor>& t, int dtype, bool split) {
    WHENET_REQUIRE(dtype == WHENET_F32 || dtype == WHENET_F16, WHENET_EINVAL, "dtype must be WHENET_F32, WHENET_F16 or WHENET_F32S");
    WHENET_REQUIRE(!split || dtype == WHENET_F32, WHENET_EINVAL, "the split-product form belongs to float32 storage");
    HostModel m;
    m.dtype = dtype;
    m.n_tensors = int(t.size());
    WHENET_REQUIRE(m.n_tensors == 315, WHENET_EFORMAT,
                   "snapshot: expected 315 tensors, found " + std::to_string(m.n_tensors));
    for (auto& kv : t) {
        const std::string head = kv.first.substr(0, kv.first.find('/'));
        if (head == "yaw" || head == "pitch" || head == "roll") m.params_heads += int64_t(kv.second.count);
        else m.params_backbone += int64_t(kv.second.count);
    }

    normalise_table(m.lut);

    {   // stem: Conv2D(32, 3x3, s2, same, no bias) + BN  -> [27][32]
        const float* w = need(t, "stem/conv/kernel", {3, 3, 3, 32}).data;
        Folded f = fold_bn(t, "stem/bn", 32);
        m.stem_w.resize(27 * 32);
        m.stem_b.resize(32);
        for (int tap = 0; tap < 27; ++tap)
            for (int co = 0; co < 32; ++co) m.stem_w[tap * 32 + co] = float(double(w[tap * 32 + co]) * f.scale[co]);
        for (int co = 0; co < 32; ++co) m.stem_b[co] = float(f.shift[co]);
    }

    for (const BlockSpec& b : make_blocks()) {
        HostBlock hb;
        hb.spec = b;
        const std::string p = "b" + std::to_string(b.index);
        const uint32_t cin = b.cin, cexp = b.cexp(), cout = b.cout, k = b.k, r = b.se_reduced();
        if (b.has_expand()) hb.expand = make_pw(t, p + "/expand", p + "/expand_bn", cin, cexp, dtype, split);
        {
            const float* w = need(t, p + "/dw/kernel", {k, k, cexp, 1}).data;
            Folded f = fold_bn(t, p + "/dw_bn", cexp);
            hb.dw.k = b.k;
            hb.dw.C = int(cexp);
            hb.dw.w.resize(size_t(k) * k * cexp);
            hb.dw.bias.resize(cexp);
            for (uint32_t tap = 0; tap < k * k; ++tap)
                for (uint32_t c = 0; c < cexp; ++c)
                    hb.dw.w[size_t(tap) * cexp + c] = float(double(w[size_t(tap) * cexp + c]) * f.scale[c]);
            for (uint32_t c = 0; c < cexp; ++c) hb.dw.bias[c] = float(f.shift[c]);
        }
        {
            const float* w1 = need(t, p + "/se_reduce/kernel", {1, 1, cexp, r}).data;   // [C][R]
            const float* b1 = need(t, p + "/se_reduce/bias", {r}).data;
            const float* w2 = need(t, p + "/se_expand/kernel", {1, 1, r, cexp}).data;   // [R][C]
            const float* b2 = need(t, p + "/se_expand/bias", {cexp}).data;
            hb.se.C = int(cexp);
            hb.se.R = int(r);
            hb.se.w1t.resize(size_t(r) * cexp);
            for (uint32_t c = 0; c < cexp; ++c)
                for (uint32_t j = 0; j < r; ++j) hb.se.w1t[size_t(j) * cexp + c] = w1[size_t(c) * r + j];
            hb.se.b1.assign(b1, b1 + r);
            hb.se.w2.assign(w2, w2 + size_t(r) * cexp);
            const uint32_t rp = (r + 3u) & ~3u;
            hb.se.w2c.assign(size_t(cexp) * rp, 0.0f);
            for (uint32_t j = 0; j < r; ++j)
                for (uint32_t c = 0; c < cexp; ++c) hb.se.w2c[size_t(c) * rp + j] = w2[size_t(j) * cexp + c];
            hb.se.b2.assign(b2, b2 + cexp);
            // round 6: the excite kernel as an MFMA operand image [ceil(R/16)][C/32][64 lanes][8] -- the project GEMMs of the 14x14 /
            // 7x7 blocks compute gate = sigmoid(r . W2 + b2) for their own k-groups on the matrix cores (pw.hip, GM = 3): binary16 for
            // f16 handles, [hi | lo] pairs for f32s; the exact-f32 configuration keeps the stand-alone kernel
            if (dtype == WHENET_F16 || split) {
                FoldedPw e;
                e.wf.resize(size_t(r) * cexp);
                for (size_t i = 0; i < e.wf.size(); ++i) e.wf[i] = double(w2[i]);
                e.shift.assign(cexp, 0.0);
                hb.se.excite = pack_pw(e, r, cexp, dtype, split);
            }
        }
        hb.project = make_pw(t, p + "/project", p + "/project_bn", cexp, cout, dtype, split);
        m.blocks.push_back(std::move(hb));
    }

    m.head = make_pw(t, "head/conv", "head/bn", 320, FEAT, dtype, split);

    {   // Block 1's project (32 -> 16, linear: BN, no activation) followed by block 2's expand (16 -> 96) is ONE affine
        // map of block 1's gated depthwise output (block 2 has no skip, nothing else reads block 1's output):
        //     expand2(project1(a)) = a . (Wp1 . We2) + (bp1 . We2 + be2),        composed here in double.
        // The engine feeds block 2's front kernel from block 1's depthwise output with this image and drops block 1's
        // project launch (option fold12); the two-step weights stay for the per-block operators and fold12 = 0.
        const BlockSpec& b1 = m.blocks[0].spec;
        const BlockSpec& b2 = m.blocks[1].spec;
        const uint32_t K = b1.cexp(), J = b1.cout, N = b2.cexp();
        WHENET_REQUIRE(!b1.has_expand() && !b2.has_skip() && uint32_t(b2.cin) == J, WHENET_EFORMAT, "fold12: unexpected block layout");
        const FoldedPw p1 = fold_pw(t, "b1/project", "b1/project_bn", K, J);
        const FoldedPw e2 = fold_pw(t, "b2/expand", "b2/expand_bn", J, N);
        FoldedPw c;
        c.wf.assign(size_t(K) * N, 0.0);
        c.shift = e2.shift;
        for (uint32_t j = 0; j < J; ++j)
            for (uint32_t n = 0; n < N; ++n) {
                const double w = e2.wf[size_t(j) * N + n];
                c.shift[n] += p1.shift[j] * w;
                for (uint32_t k = 0; k < K; ++k) c.wf[size_t(k) * N + n] += p1.wf[size_t(k) * J + j] * w;
            }
        m.fold12 = pack_pw(c, K, N, dtype, split);
        // the f32 image front2.hip scales by the crop's gate before rounding: the f16 fragment order (16 k per step,
        // lane l <-> n = 32 ntile + (l & 31), k = 16 ks + 8 (l >> 5) + e), whatever the handle's dtype
        const int KS = ceil_div(int(K), 16), NT = ceil_div(int(N), 32);
        m.fold12_w32.assign(size_t(KS) * NT * 64 * 8, 0.0f);
        for (int ks = 0; ks < KS; ++ks)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t n = uint32_t(nt * 32 + (lane & 31)), k = uint32_t(ks * 16 + (lane >> 5) * 8 + e);
                        if (n < N && k < K)
                            m.fold12_w32[((size_t(ks) * NT + nt) * 64 + lane) * 8 + e] = float(c.wf[size_t(k) * N + n]);
                    }
    }

    m.dense_w.resize(size_t(FEAT) * N_LOGITS);
    m.dense_b.resize(N_LOGITS);
    int col = 0;
    const char* names[3] = {"yaw", "pitch", "roll"};
    const uint32_t widths[3] = {N_YAW, N_PITCH, N_ROLL};
    for (int h = 0; h < 3; ++h) {
        const float* w = need(t, std::string(names[h]) + "/kernel", {FEAT, widths[h]}).data;
        const float* b = need(t, std::string(names[h]) + "/bias", {widths[h]}).data;
        for (int i = 0; i < FEAT; ++i)
            for (uint32_t j = 0; j < widths[h]; ++j) m.dense_w[size_t(i) * N_LOGITS + col + j] = w[size_t(i) * widths[h] + j];
        for (uint32_t j = 0; j < widths[h]; ++j) m.dense_b[col + j] = b[j];
        col += int(widths[h]);
    }
    WHENET_REQUIRE(m.params_backbone == 4049564 && m.params_heads == 322812, WHENET_EFORMAT,
                   "snapshot: parameter census mismatch");
    return m;
}

}  // namespace whenet

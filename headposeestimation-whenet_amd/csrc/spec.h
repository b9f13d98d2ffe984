// EfficientNet-B0 (no top) block table, C++ side.  Independent copy of
// whenet_hip/spec.py (tests compare the two through whenet_block_spec()).
// Follows efficientnet==0.0.4 params.py block strings for B0, which is what
// /root/reference/whenet.py:8 instantiates.
#pragma once

#include <array>
#include <string>
#include <vector>

namespace whenet {

struct BlockSpec {
    int index;       // 1..16
    int k, s, expand, cin, cout, h_in, h_out;
    int cexp() const { return cin * expand; }
    int se_reduced() const { int r = int(cin * 0.25); return r < 1 ? 1 : r; }  // on INPUT filters
    bool has_expand() const { return expand != 1; }
    bool has_skip() const { return s == 1 && cin == cout; }
    // TF 'SAME': pad_before for the depthwise conv
    int pad_before() const {
        int total = (h_out - 1) * s + k - h_in;
        if (total < 0) total = 0;
        return total / 2;
    }
};

inline std::vector<BlockSpec> make_blocks() {
    // (repeats, kernel, stride, expand, in, out)
    static const int stages[7][6] = {
        {1, 3, 1, 1, 32, 16},  {2, 3, 2, 6, 16, 24},   {2, 5, 2, 6, 24, 40},  {3, 3, 2, 6, 40, 80},
        {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};
    std::vector<BlockSpec> out;
    int h = 112, idx = 0;
    for (auto& st : stages) {
        for (int j = 0; j < st[0]; ++j) {
            BlockSpec b;
            b.index = ++idx;
            b.k = st[1];
            b.s = (j == 0) ? st[2] : 1;
            b.expand = st[3];
            b.cin = (j == 0) ? st[4] : st[5];
            b.cout = st[5];
            b.h_in = h;
            b.h_out = (h + b.s - 1) / b.s;
            h = b.h_out;
            out.push_back(b);
        }
    }
    return out;
}

}  // namespace whenet

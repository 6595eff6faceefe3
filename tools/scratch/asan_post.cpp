#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <random>
extern "C" int sd_nw_identity_batch(const char* const*, const int32_t*, const char* const*, const int32_t*, int64_t, int32_t, int32_t*, int32_t*, int32_t*);
extern "C" int sd_identity_segments(const char*, int64_t, const int64_t*, const int64_t*, int64_t, const char* const*, const int32_t*, int32_t, const int32_t*, int32_t, int32_t, int32_t*, int32_t*, int32_t*);
int main() {
    std::mt19937 g(7);
    auto rnd = [&](int n, int k) { std::string s(n, 'A'); for (auto& c : s) c = "ACGT"[g() % k]; return s; };
    std::vector<std::string> q, t;
    for (int i = 0; i < 12; ++i) { t.push_back(rnd(i % 3 == 0 ? 171 : i % 3 == 1 ? 600 : 2000, i % 4 ? 4 : 2)); q.push_back(rnd(i % 3 == 0 ? 20000 + i : i % 3 == 1 ? 6000 + i : 3000 + i, 4)); }
    q.push_back(""); t.push_back("ACGT"); q.push_back("ACGT"); t.push_back(""); q.push_back(std::string(100000, 'A')); t.push_back(std::string(170, 'A') + "C");
    q.push_back(std::string("\x01\x02\xff""ACGT", 7)); t.push_back(std::string("\xff\x02\x01TGCA", 7));
    std::string all; for (int c = 1; c < 256; ++c) all.push_back((char)c); q.push_back(all + all); t.push_back(all);
    std::vector<const char*> qp, tp; std::vector<int32_t> ql, tl;
    for (size_t i = 0; i < q.size(); ++i) { qp.push_back(q[i].data()); tp.push_back(t[i].data()); ql.push_back((int32_t)q[i].size()); tl.push_back((int32_t)t[i].size()); }
    std::vector<int32_t> d(q.size()), m(q.size()), c(q.size());
    int rc = sd_nw_identity_batch(qp.data(), ql.data(), tp.data(), tl.data(), (int64_t)q.size(), 4, d.data(), m.data(), c.data());
    long sum = 0; for (size_t i = 0; i < q.size(); ++i) sum += d[i] * 3 + m[i];
    std::printf("rc %d checksum %ld\n", rc, sum);
    // segments form, homopolymer compression
    std::string seq = rnd(50000, 4) + std::string(500, 'N') + rnd(3000, 2);
    std::vector<int64_t> st{0, 100, 49000, 30000}, en{170, 25000, 53499, 30000};
    std::vector<const char*> tt{t[0].data(), t[1].data()}; std::vector<int32_t> ttl{(int32_t)t[0].size(), (int32_t)t[1].size()};
    std::vector<int32_t> d2(8), m2(8), c2(8);
    for (int homo = 0; homo < 2; ++homo) {
        rc = sd_identity_segments(seq.data(), (int64_t)seq.size(), st.data(), en.data(), 4, tt.data(), ttl.data(), 2, nullptr, homo, 3, d2.data(), m2.data(), c2.data());
        std::printf("segments homo %d rc %d first %d %d %d\n", homo, rc, d2[0], m2[0], c2[0]);
    }
    return 0;
}

// C++ host-side check through wfa_amd/host/wfa.hpp: mirrors the reference's own test (wfa_test.go:30-185 builds an
// aligner with adaptive 10/50/1, aligns a pair, prints CIGAR / region / stats) but ASSERTS the README's
// published outputs.  Exit code 0 = all good; 77 = no GPU (skipped).
#include <cstdio>
#include <thread>
#include <tuple>
#include "../wfa_amd/host/wfa.hpp"

static int fails = 0;
#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);     \
            fails++;                                                    \
        }                                                               \
    } while (0)

int main() {
    if (wfahip_device_count() <= 0) {
        std::printf("no HIP device: skipped\n");
        return 77;
    }
    {
        auto algn = wfa::New(wfa::Penalties{4, 6, 2}, wfa::Options{true});
        CHECK(algn->ok());
        CHECK(algn->AdaptiveReduction(wfa::AdaptiveReductionOption{0, 50, 1}) == wfa::Error::BadAdaptiveOption);
        CHECK(algn->AdaptiveReduction(wfa::AdaptiveReductionOption{10, 50, 1}) == wfa::Error::None);
        wfa::Error err;
        auto       r = algn->Align("ACCATACTCG", "AGGATGCTCG", &err);  // README.md:101-124
        CHECK(err == wfa::Error::None);
        CHECK(r.CIGAR(false) == "1M2X2M1X4M");
        CHECK(r.Score == 12 && r.QBegin == 1 && r.QEnd == 10 && r.TBegin == 1 && r.TEnd == 10);
        CHECK(r.AlignLen == 10 && r.Matches == 7 && r.Gaps == 0 && r.GapRegions == 0);
        r = algn->Align("AGCTAGTGTCAATGGCTACTTTTCAGGTCCT", "AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT", &err);  // README.md:231-239
        CHECK(r.CIGAR(false) == "1M1X3M1I5M2X8M3I1M1X9M" && r.Score == 36);
        std::string Q, A, T;
        r.AlignmentText("AGCTAGTGTCAATGGCTACTTTTCAGGTCCT", "AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT", false, Q, A, T);
        CHECK(Q == "AGCTA-GTGTCAATGGCTACT---TTTCAGGTCCT");
        CHECK(A == "| ||| |||||  ||||||||   | |||||||||");
        CHECK(T == "AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT");
        algn->Align("", "ACGT", &err);
        CHECK(err == wfa::Error::EmptySeq);
        wfa::RecycleAligner(algn);
    }
    {
        auto algn = wfa::New(wfa::DefaultPenalties, wfa::Options{false});
        algn->AdaptiveReduction(wfa::DefaultAdaptiveOption);
        wfa::Error err;
        auto r = algn->Align("Bioinformatics helps Biology", "We learn bioinformatics to help biologists", &err);  // README.md:18-27
        CHECK(r.CIGAR(false) == "9I1X14M3I4M1D1M1X5M1X3I" && r.Score == 32);
        CHECK(r.CIGAR(true) == "14M3I4M1D1M1X5M");
        CHECK(r.QBegin == 2 && r.QEnd == 27 && r.TBegin == 11 && r.TEnd == 38);
        CHECK(r.AlignLen == 29 && r.Matches == 24 && r.Gaps == 4 && r.GapRegions == 2);
        std::vector<wfa::AlignmentResult> rs;
        std::vector<wfa::Error>           es;
        algn->AlignBatch({"ACGATCTCG", "", "A"}, {"CAGGCTCCTCGG", "A", "CA"}, rs, es);
        CHECK(es[0] == wfa::Error::None && es[1] == wfa::Error::EmptySeq && es[2] == wfa::Error::None);
        CHECK(rs[0].Score == 16 && rs[0].QBegin == 1 && rs[0].QEnd == 9 && rs[0].TBegin == 2 && rs[0].TEnd == 11);
    }
    {
        // one aligner per thread (the reference's rule, wfa.go:73-78): two contexts on the same GPU, used
        // concurrently, must give what one context gives; the same batch through a context SET (the multi-GPU
        // entry, here two contexts on device 0) comes back in pair order
        std::vector<std::string> qs, ts;
        uint64_t                 x = 88172645463325252ull;
        auto                     rnd = [&] { x ^= x << 13, x ^= x >> 7, x ^= x << 17; return x; };
        for (int i = 0; i < 3000; i++) {
            std::string q(40 + rnd() % 400, 'A');
            for (auto &c : q) c = "ACGT"[rnd() % 4];
            std::string t = q;
            for (int e = 0; e < (int)(q.size() / 20); e++) {
                const size_t pos = rnd() % t.size();
                switch (rnd() % 3) {
                case 0: t[pos] = "ACGT"[rnd() % 4]; break;
                case 1: t.insert(pos, 1, "ACGT"[rnd() % 4]); break;
                default: if (t.size() > 1) t.erase(pos, 1);
                }
            }
            if (i % 500 == 17) q.clear();  // ErrEmptySeq inside a shard
            qs.push_back(q), ts.push_back(t);
        }
        auto one = wfa::New();
        one->AdaptiveReduction(wfa::DefaultAdaptiveOption);
        std::vector<wfa::AlignmentResult> want, got[2], gm;
        std::vector<wfa::Error>           ew, eg[2], em;
        CHECK(one->AlignBatch(qs, ts, want, ew) == 0);
        std::thread th[2];
        for (int k = 0; k < 2; k++)
            th[k] = std::thread([&, k] {
                auto a = wfa::New();
                a->AdaptiveReduction(wfa::DefaultAdaptiveOption);
                for (int rep = 0; rep < 3; rep++) a->AlignBatch(qs, ts, got[k], eg[k]);
            });
        for (auto &t : th) t.join();
        const int     ids[2] = {0, 0};
        wfahip_multi *m      = nullptr;
        CHECK(wfahip_create_multi(ids, 2, &m) == 0 && wfahip_multi_size(m) == 2);
        CHECK(one->AlignBatch(qs, ts, gm, em, m) == 0);
        wfahip_destroy_multi(m);
        for (size_t i = 0; i < qs.size(); i++) {
            CHECK(ew[i] == (qs[i].empty() ? wfa::Error::EmptySeq : wfa::Error::None));
            for (int k = 0; k < 2; k++) CHECK(eg[k][i] == ew[i] && got[k][i].Ops == want[i].Ops && got[k][i].Score == want[i].Score);
            CHECK(em[i] == ew[i] && gm[i].Ops == want[i].Ops && gm[i].Score == want[i].Score && gm[i].TEnd == want[i].TEnd);
        }
        // per-pair submissions collected as one batch
        for (size_t i = 0; i < 200; i++) {
            uint64_t tk = ~0ull;
            CHECK(one->Submit(qs[i], ts[i], &tk) && tk == i);
        }
        std::vector<wfa::AlignmentResult> gs;
        std::vector<wfa::Error>           es2;
        CHECK(one->Collect(gs, es2) == 0 && gs.size() == 200);
        for (size_t i = 0; i < gs.size(); i++) CHECK(es2[i] == ew[i] && gs[i].Ops == want[i].Ops);
        CHECK(one->Collect(gs, es2) == 0 && gs.empty());
    }
    std::printf(fails ? "%d checks failed\n" : "cpp host test ok\n", fails);
    return fails ? 1 : 0;
}

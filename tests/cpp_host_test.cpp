// C++ host-side check through wfa_amd/host/wfa.hpp: mirrors the reference's own test (wfa_test.go:30-185 builds an
// aligner with adaptive 10/50/1, aligns a pair, prints CIGAR / region / stats) but ASSERTS the README's
// published outputs.  Exit code 0 = all good; 77 = no GPU (skipped).
#include <cstdio>
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
    std::printf(fails ? "%d checks failed\n" : "cpp host test ok\n", fails);
    return fails ? 1 : 0;
}

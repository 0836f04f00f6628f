// wfa.hpp -- C++ host-side mirror of the reference's Go API (package wfa) above the libwfahip.so C-ABI.
//
// The reference is compiled Go and no Go toolchain exists in the build image, so the compiled host side is
// written in C++ (header-only, links only against libwfahip.so).  Names, argument meaning and error
// behaviour follow the reference (file:line into shenwei356/wfa v0.4.0):
//
//   wfa::Penalties / DefaultPenalties                wfa.go:32-43
//   wfa::AdaptiveReductionOption / DefaultAdaptive   wfa.go:46-60
//   wfa::Options / DefaultOptions                    wfa.go:64-71
//   wfa::New, RecycleAligner                         wfa.go:102-131
//   Aligner::AdaptiveReduction / Align               wfa.go:134-140,196-268
//   ErrEmptySeq / ErrSeqTooLong / MaxSeqLen          wfa.go:186-193
//   AlignmentResult, Op, CIGAR, AlignmentText        wfa_cigar.go:30-66,236-333
//
// Go returns (value, error); here Align returns the result and reports the error through an Error code
// (no exceptions cross the API).  AlignBatch is new: the batch entry a GPU needs.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wfa_hip.h"

namespace wfa {

struct Penalties {
    uint32_t Mismatch = 4, GapOpen = 6, GapExt = 2;
};
struct AdaptiveReductionOption {
    uint32_t MinWFLen = 10, MaxDistDiff = 50, CutoffStep = 1;
};
struct Options {
    bool GlobalAlignment = true;
};
inline const Penalties               DefaultPenalties{};
inline const AdaptiveReductionOption DefaultAdaptiveOption{};
inline const Options                 DefaultOptions{};

constexpr int      MaxSeqLen   = (1 << 29) - 1;
constexpr uint64_t MaskLower32 = 4294967295ull;
constexpr uint64_t OpM = 'M', OpD = 'D', OpI = 'I', OpX = 'X', OpH = 'H';

enum class Error { None = 0, EmptySeq, SeqTooLong, BadAdaptiveOption, NoMemory, Device };
inline const char *ErrorText(Error e) {
    switch (e) {
    case Error::None: return "";
    case Error::EmptySeq: return "wfa: invalid empty sequence";                                // wfa.go:187
    case Error::SeqTooLong: return "wfa: sequences longer than 536870911 are not supported";   // wfa.go:193
    case Error::BadAdaptiveOption: return "cutoff step should not be 0";                       // wfa.go:136
    case Error::NoMemory: return "wfa: out of device memory for this pair";
    case Error::Device: return "wfa: device error";
    }
    return "";
}

inline std::pair<char, uint32_t> Op(uint64_t op) { return {(char)(op >> 32), (uint32_t)(op & MaskLower32)}; }

struct AlignmentResult {
    std::vector<uint64_t> Ops;  // op<<32 | n, reversed + merged (process(), wfa_cigar.go:136-214)
    uint32_t              Score = 0;
    int                   TBegin = 0, TEnd = 0, QBegin = 0, QEnd = 0;
    uint32_t              AlignLen = 0, Matches = 0, Gaps = 0, GapRegions = 0;

    // ops between the first and the last M run (trimOps, wfa_cigar.go:217-233)
    std::pair<size_t, size_t> AlignedSpan() const {
        size_t first = Ops.size(), last = 0;
        bool   any = false;
        for (size_t i = 0; i < Ops.size(); i++)
            if ((Ops[i] >> 32) == OpM) {
                if (!any) first = i;
                last = i, any = true;
            }
        return any ? std::make_pair(first, last + 1) : std::make_pair((size_t)0, (size_t)0);
    }
    std::string CIGAR(bool onlyAlignedRegion) const {  // wfa_cigar.go:236-255
        size_t b = 0, e = Ops.size();
        if (onlyAlignedRegion) std::tie(b, e) = AlignedSpan();
        std::string s;
        for (size_t i = b; i < e; i++) {
            s += std::to_string((uint32_t)(Ops[i] & MaskLower32));
            s += (char)(Ops[i] >> 32);
        }
        return s;
    }
    // the three display lines: query, match bars, target (wfa_cigar.go:259-333)
    void AlignmentText(const std::string &q0, const std::string &t0, bool onlyAlignedRegion, std::string &Q,
                       std::string &A, std::string &T) const {
        size_t      b = 0, e = Ops.size();
        std::string q = q0, t = t0;
        if (onlyAlignedRegion) {
            std::tie(b, e) = AlignedSpan();
            q = q0.substr(QBegin - 1, QEnd - QBegin + 1);
            t = t0.substr(TBegin - 1, TEnd - TBegin + 1);
        }
        Q.clear(), A.clear(), T.clear();
        size_t v = 0, h = 0;
        for (size_t i = b; i < e; i++) {
            const uint64_t letter = Ops[i] >> 32;
            const size_t   n      = (size_t)(Ops[i] & MaskLower32);
            if (letter == OpM || letter == OpX) {
                Q.append(q, v, n), A.append(n, letter == OpM ? '|' : ' '), T.append(t, h, n);
                v += n, h += n;
            } else if (letter == OpI) {
                Q.append(n, '-'), A.append(n, ' '), T.append(t, h, n);
                h += n;
            } else if (letter == OpD || letter == OpH) {
                Q.append(q, v, n), A.append(n, ' '), T.append(n, '-');
                v += n;
            }
        }
    }
};

class Aligner {
  public:
    Aligner(const Penalties &p, const Options &opt, int device = -1) : p_(p), opt_(opt) {
        create_rc_ = wfahip_create(device, &ctx_);
    }
    ~Aligner() {
        if (ctx_) wfahip_destroy(ctx_);
    }
    Aligner(const Aligner &)            = delete;
    Aligner &operator=(const Aligner &) = delete;

    bool ok() const { return ctx_ != nullptr; }
    int  create_code() const { return create_rc_; }

    Error AdaptiveReduction(const AdaptiveReductionOption &ad) {  // wfa.go:134-140
        if (ad.MinWFLen == 0) return Error::BadAdaptiveOption;
        ad_ = ad, has_ad_ = true;
        return Error::None;
    }

    // wfa.go:196.  On error the returned result is empty and *err says why.
    AlignmentResult Align(const std::string &q, const std::string &t, Error *err = nullptr) {
        std::vector<AlignmentResult> rs;
        std::vector<Error>           es;
        if (q.empty() || t.empty()) {
            if (err) *err = Error::EmptySeq;
            return {};
        }
        if (!ctx_) {
            if (err) *err = Error::Device;
            return {};
        }
        // wfahip_align_pair: record + ops into reusable buffers, two launches and no copy when the pair's shape allows it
        (void)rs, (void)es;
        uint32_t rec[WFAHIP_REC_WORDS];
        uint64_t n_ops = 0;
        if (one_ops_.size() < q.size() + t.size() + 2) one_ops_.resize(2 * (q.size() + t.size()) + 64);
        const wfahip_params prm = params();
        const int rc = wfahip_align_pair(ctx_, &prm, reinterpret_cast<const uint8_t *>(q.data()), (uint32_t)q.size(),
                                         reinterpret_cast<const uint8_t *>(t.data()), (uint32_t)t.size(), rec, one_ops_.data(),
                                         one_ops_.size(), &n_ops);
        Error e = rc != WFAHIP_OK ? Error::Device
                  : rec[WFAHIP_REC_STATUS] == WFAHIP_PAIR_OK       ? Error::None
                  : rec[WFAHIP_REC_STATUS] == WFAHIP_PAIR_EMPTY    ? Error::EmptySeq
                  : rec[WFAHIP_REC_STATUS] == WFAHIP_PAIR_TOO_LONG ? Error::SeqTooLong
                                                                   : Error::NoMemory;
        if (err) *err = e;
        AlignmentResult r;
        if (e != Error::None) return r;
        r.Ops.assign(one_ops_.begin(), one_ops_.begin() + (long)n_ops);
        r.Score = rec[WFAHIP_REC_SCORE], r.TBegin = (int)rec[WFAHIP_REC_TBEGIN], r.TEnd = (int)rec[WFAHIP_REC_TEND];
        r.QBegin = (int)rec[WFAHIP_REC_QBEGIN], r.QEnd = (int)rec[WFAHIP_REC_QEND], r.AlignLen = rec[WFAHIP_REC_ALIGN_LEN];
        r.Matches = rec[WFAHIP_REC_MATCHES], r.Gaps = rec[WFAHIP_REC_GAPS], r.GapRegions = rec[WFAHIP_REC_GAP_REGIONS];
        return r;
    }

    // new: hand in one pair (copied), get its ticket; Collect aligns everything submitted so far as ONE batch and
    // returns results[ticket] / errors[ticket].  Serves a per-pair loop like wfa-go/wfa-go.go:166-178 at batch speed.
    // Returns false (and leaves *ticket alone) when the pair could not be queued -- no context, out of host memory --
    // so that the tickets of later pairs still line up with Collect's results.
    bool Submit(const std::string &q, const std::string &t, uint64_t *ticket) {
        uint64_t tk = 0;
        if (!ctx_ || wfahip_submit(ctx_, reinterpret_cast<const uint8_t *>(q.data()), (uint32_t)q.size(),
                                   reinterpret_cast<const uint8_t *>(t.data()), (uint32_t)t.size(), &tk) != WFAHIP_OK)
            return false;
        if (ticket) *ticket = tk;
        return true;
    }
    int Collect(std::vector<AlignmentResult> &results, std::vector<Error> &errors) {
        const size_t n = ctx_ ? (size_t)wfahip_pending(ctx_) : 0;
        results.assign(n, {});
        errors.assign(n, Error::None);
        if (!ctx_) return create_rc_;
        const wfahip_params prm = params();
        wfahip_results      out{};
        const int           rc = wfahip_collect(ctx_, &prm, &out);
        return unpack(rc, out, results, errors);
    }

    // new: one device call for many pairs; results[i] / errors[i] per pair.  With `multi` (a context set over several
    // GPUs, wfahip_create_multi) the batch is sharded over them.
    int AlignBatch(const std::vector<std::string> &qs, const std::vector<std::string> &ts,
                   std::vector<AlignmentResult> &results, std::vector<Error> &errors, wfahip_multi *multi = nullptr) {
        const size_t n = qs.size();
        results.assign(n, {});
        errors.assign(n, Error::None);
        if (!ctx_ && !multi) {
            errors.assign(n, Error::Device);
            return create_rc_;
        }
        if (n == 0) return 0;
        std::vector<uint64_t> q_off(n), t_off(n);
        std::vector<uint32_t> q_len(n), t_len(n);
        uint64_t              total = 0;
        for (size_t i = 0; i < n; i++) {  // 16-byte aligned starts
            q_off[i] = total, q_len[i] = (uint32_t)qs[i].size(), total += (qs[i].size() + 15) & ~(size_t)15;
            t_off[i] = total, t_len[i] = (uint32_t)ts[i].size(), total += (ts[i].size() + 15) & ~(size_t)15;
        }
        std::vector<uint8_t> blob(total + 16, 0);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(blob.data() + q_off[i], qs[i].data(), qs[i].size());
            std::memcpy(blob.data() + t_off[i], ts[i].data(), ts[i].size());
        }
        const wfahip_params prm = params();
        wfahip_results      out{};
        const int rc = multi ? wfahip_align_batch_multi(multi, &prm, blob.data(), blob.size(), q_off.data(), q_len.data(),
                                                        t_off.data(), t_len.data(), n, &out)
                             : wfahip_align_batch(ctx_, &prm, blob.data(), blob.size(), q_off.data(), q_len.data(),
                                                  t_off.data(), t_len.data(), n, &out);
        return unpack(rc, out, results, errors);
    }

  private:
    wfahip_params params() const {
        wfahip_params prm{};
        prm.mismatch = p_.Mismatch, prm.gap_open = p_.GapOpen, prm.gap_ext = p_.GapExt;
        prm.global_alignment = opt_.GlobalAlignment ? 1 : 0;
        if (has_ad_) {
            prm.adaptive   = 1;
            prm.min_wf_len = ad_.MinWFLen, prm.max_dist_diff = ad_.MaxDistDiff, prm.cutoff_step = ad_.CutoffStep;
        }
        return prm;
    }
    static int unpack(int rc, wfahip_results &out, std::vector<AlignmentResult> &results, std::vector<Error> &errors) {
        const size_t n = results.size();
        if (rc != 0) {
            errors.assign(n, Error::Device);
            return rc;
        }
        for (size_t i = 0; i < n; i++) {
            switch (out.status[i]) {
            case WFAHIP_PAIR_OK: {
                AlignmentResult &r = results[i];
                r.Ops.assign(out.ops + out.ops_off[i], out.ops + out.ops_off[i] + out.ops_len[i]);
                r.Score = out.score[i];
                r.TBegin = out.tbegin[i], r.TEnd = out.tend[i], r.QBegin = out.qbegin[i], r.QEnd = out.qend[i];
                r.AlignLen = out.align_len[i], r.Matches = out.matches[i], r.Gaps = out.gaps[i];
                r.GapRegions = out.gap_regions[i];
                break;
            }
            case WFAHIP_PAIR_EMPTY: errors[i] = Error::EmptySeq; break;
            case WFAHIP_PAIR_TOO_LONG: errors[i] = Error::SeqTooLong; break;
            default: errors[i] = Error::NoMemory; break;
            }
        }
        wfahip_results_free(&out);
        return 0;
    }

    Penalties               p_;
    Options                 opt_;
    AdaptiveReductionOption ad_{};
    bool                    has_ad_ = false;
    wfahip_ctx             *ctx_    = nullptr;
    std::vector<uint64_t>   one_ops_;  // Align's reusable CIGAR buffer
    int                     create_rc_ = 0;
};

inline std::unique_ptr<Aligner> New(const Penalties &p = DefaultPenalties, const Options &opt = DefaultOptions,
                                    int device = -1) {
    return std::make_unique<Aligner>(p, opt, device);
}
inline void RecycleAligner(std::unique_ptr<Aligner> &a) { a.reset(); }  // wfa.go:102
inline void RecycleAlignmentResult(AlignmentResult &) {}                // wfa_cigar.go:92

}  // namespace wfa

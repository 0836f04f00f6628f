// Package wfa is the drop-in replacement of github.com/shenwei356/wfa's alignment path: the same exported
// API (Penalties, Options, AdaptiveReductionOption, New, RecycleAligner, (*Aligner).AdaptiveReduction,
// Align, AlignPointers, AlignmentResult, CIGAR, ...), with the wavefront extend/next loop, wf-adaptive
// reduction and backtrace running on an AMD MI355X through libwfahip.so (include/wfa_hip.h).
//
// NOTE: the build container has no Go toolchain, so this file has never been compiled; it is kept
// deliberately thin (buffer marshalling only) and mirrors wfa_amd/aligner.py, which IS exercised by the
// test-suite through the identical C-ABI.  See INTEGRATION.md.
package wfa

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../wfa_amd/lib -lwfahip
#include <stdlib.h>
#include "wfa_hip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

// Penalties contains the gap-affine penalties, Match is 0.  (reference: wfa.go:32-36)
type Penalties struct {
	Mismatch uint32
	GapOpen  uint32
	GapExt   uint32
}

// DefaultPenalties is from the WFA paper.  (wfa.go:39-43)
var DefaultPenalties = &Penalties{Mismatch: 4, GapOpen: 6, GapExt: 2}

// AdaptiveReductionOption contains the parameters for adaptive reduction.  (wfa.go:46-50)
type AdaptiveReductionOption struct {
	MinWFLen    uint32
	MaxDistDiff uint32
	CutoffStep  uint32 // not used yet (by the reference either).
}

// DefaultAdaptiveOption: 10, 50, 1.  (wfa.go:56-60)
var DefaultAdaptiveOption = &AdaptiveReductionOption{MinWFLen: 10, MaxDistDiff: 50, CutoffStep: 1}

// Options: global or semi-global alignment.  (wfa.go:64-66)
type Options struct {
	GlobalAlignment bool
}

// DefaultOptions is the default option.  (wfa.go:69-71)
var DefaultOptions = &Options{GlobalAlignment: true}

// ErrEmptySeq / ErrSeqTooLong / MaxSeqLen: wfa.go:186-193.
var ErrEmptySeq error = fmt.Errorf("wfa: invalid empty sequence")

const MaxSeqLen int = 1<<(32-3) - 1

// The C header's enum constants as TYPED Go constants.  cgo hands enum constants over as constants whose type depends on
// the toolchain (untyped integers with some versions, the enum's C type with others); array lengths, indices and `case`
// labels below use these conversions, which are constant expressions either way.
const (
	recWords      = int(C.WFAHIP_REC_WORDS)
	recStatus     = int(C.WFAHIP_REC_STATUS)
	recScore      = int(C.WFAHIP_REC_SCORE)
	recTBegin     = int(C.WFAHIP_REC_TBEGIN)
	recTEnd       = int(C.WFAHIP_REC_TEND)
	recQBegin     = int(C.WFAHIP_REC_QBEGIN)
	recQEnd       = int(C.WFAHIP_REC_QEND)
	recAlignLen   = int(C.WFAHIP_REC_ALIGN_LEN)
	recMatches    = int(C.WFAHIP_REC_MATCHES)
	recGaps       = int(C.WFAHIP_REC_GAPS)
	recGapRegions = int(C.WFAHIP_REC_GAP_REGIONS)
	recOpsLen     = int(C.WFAHIP_REC_OPS_LEN)
	pairOK        = uint32(C.WFAHIP_PAIR_OK)
	pairEmpty     = uint32(C.WFAHIP_PAIR_EMPTY)
	pairTooLong   = uint32(C.WFAHIP_PAIR_TOO_LONG)
)

var ErrSeqTooLong error = fmt.Errorf("wfa: sequences longer than %d are not supported", MaxSeqLen)

// Aligner holds one device context.  Like the reference's (wfa.go:73-78) it must not be used from several
// goroutines at once; create one per goroutine (different Aligners may run concurrently).
type Aligner struct {
	p     *Penalties
	ad    *AdaptiveReductionOption
	opt   *Options
	ctx   *C.wfahip_ctx
	multi *C.wfahip_multi // NewMulti: AlignBatch shards over several GPUs
	oneOps []C.uint64_t   // Align's reusable CIGAR buffer (wfahip_align_pair)

	// M, I, D: the reference exports its three components (wfa.go:86) for Plot and its test (wfa_test.go:154).
	// On the GPU path the wavefronts live in HBM and are released per batch; Plot below fetches one pair's
	// stored rows on demand (wfahip_debug_wavefronts) instead of keeping these fields.
}

// New returns a new Aligner bound to the current HIP device.  (wfa.go:120)
func New(p *Penalties, opt *Options) *Aligner {
	algn := &Aligner{p: p, opt: opt}
	if rc := C.wfahip_create(C.int(-1), &algn.ctx); rc != 0 {
		panic(fmt.Sprintf("wfa: %s", C.GoString(C.wfahip_strerror(rc))))
	}
	return algn
}

// NewMulti returns an Aligner whose AlignBatch shards the pairs over the given GPUs (nil = every GPU): one context
// and one host thread per GPU inside the library, results merged in pair order.  The reference's model is one
// Aligner per goroutine (wfa.go:73-78); this is that model behind one call.  Align / Submit use the first GPU.
func NewMulti(p *Penalties, opt *Options, devices []int) *Aligner {
	algn := &Aligner{p: p, opt: opt}
	var ids *C.int
	cids := make([]C.int, len(devices))
	for i, d := range devices {
		cids[i] = C.int(d)
	}
	if len(cids) > 0 {
		ids = &cids[0]
	}
	if rc := C.wfahip_create_multi(ids, C.int(len(cids)), &algn.multi); rc != 0 {
		panic(fmt.Sprintf("wfa: %s", C.GoString(C.wfahip_strerror(rc))))
	}
	algn.ctx = C.wfahip_multi_ctx(algn.multi, 0) // owned by the set
	return algn
}

// RecycleAligner releases the device context(s).  (wfa.go:102)
func RecycleAligner(algn *Aligner) {
	if algn == nil {
		return
	}
	if algn.multi != nil {
		C.wfahip_destroy_multi(algn.multi)
		algn.multi, algn.ctx = nil, nil
	} else if algn.ctx != nil {
		C.wfahip_destroy(algn.ctx)
		algn.ctx = nil
	}
}

// AdaptiveReduction sets the adaptive reduction parameters.  (wfa.go:134-140, same error text)
func (algn *Aligner) AdaptiveReduction(ad *AdaptiveReductionOption) error {
	if ad.MinWFLen == 0 {
		return fmt.Errorf("cutoff step should not be 0")
	}
	algn.ad = ad
	return nil
}

func (algn *Aligner) params() C.wfahip_params {
	var p C.wfahip_params
	p.mismatch, p.gap_open, p.gap_ext = C.uint32_t(algn.p.Mismatch), C.uint32_t(algn.p.GapOpen), C.uint32_t(algn.p.GapExt)
	if algn.opt.GlobalAlignment {
		p.global_alignment = 1
	}
	if algn.ad != nil {
		p.adaptive = 1
		p.min_wf_len, p.max_dist_diff, p.cutoff_step = C.uint32_t(algn.ad.MinWFLen), C.uint32_t(algn.ad.MaxDistDiff), C.uint32_t(algn.ad.CutoffStep)
	}
	return p
}

// Align performs alignment with two sequences.  (wfa.go:196)
func (algn *Aligner) Align(q, t []byte) (*AlignmentResult, error) {
	return algn.AlignPointers(&q, &t)
}

// AlignPointers performs alignment with two sequences. The arguments are pointers.  (wfa.go:201)
func (algn *Aligner) AlignPointers(q, t *[]byte) (*AlignmentResult, error) {
	if len(*q) == 0 || len(*t) == 0 {
		return nil, ErrEmptySeq
	}
	if len(*q) > MaxSeqLen || len(*t) > MaxSeqLen {
		return nil, ErrSeqTooLong
	}
	if algn.multi != nil { // (a context set: the batch entry)
		rs, errs := algn.AlignBatch([][]byte{*q}, [][]byte{*t})
		return rs[0], errs[0]
	}
	// wfahip_align_pair: the record and the ops come back in two reusable buffers -- no offset arrays to build, no
	// result arrays to take apart, and (for pairs shaped like the reference's defaults) two launches and no copy.
	need := len(*q) + len(*t) + 2
	if len(algn.oneOps) < need {
		algn.oneOps = make([]C.uint64_t, 2*need+64)
	}
	var rec [recWords]C.uint32_t
	var nOps C.uint64_t
	p := algn.params()
	rc := C.wfahip_align_pair(algn.ctx, &p, (*C.uint8_t)(unsafe.Pointer(&(*q)[0])), C.uint32_t(len(*q)),
		(*C.uint8_t)(unsafe.Pointer(&(*t)[0])), C.uint32_t(len(*t)), &rec[0], &algn.oneOps[0], C.uint64_t(len(algn.oneOps)), &nOps)
	runtime.KeepAlive(q)
	runtime.KeepAlive(t)
	if rc != 0 {
		return nil, fmt.Errorf("wfa: %s", C.GoString(C.wfahip_strerror(rc)))
	}
	switch uint32(rec[recStatus]) {
	case pairEmpty:
		return nil, ErrEmptySeq
	case pairTooLong:
		return nil, ErrSeqTooLong
	case pairOK:
	default:
		return nil, fmt.Errorf("wfa: not enough device memory for this pair")
	}
	r := NewAlignmentResult(algn.opt.GlobalAlignment)
	r.proccessed = true // ops arrive reversed + merged (process(), wfa_cigar.go:136-214, ran on the device)
	r.Ops = r.Ops[:0]
	for i := 0; i < int(nOps); i++ {
		r.Ops = append(r.Ops, uint64(algn.oneOps[i]))
	}
	r.Score = uint32(rec[recScore])
	r.TBegin, r.TEnd = int(int32(rec[recTBegin])), int(int32(rec[recTEnd]))
	r.QBegin, r.QEnd = int(int32(rec[recQBegin])), int(int32(rec[recQEnd]))
	r.AlignLen, r.Matches = uint32(rec[recAlignLen]), uint32(rec[recMatches])
	r.Gaps, r.GapRegions = uint32(rec[recGaps]), uint32(rec[recGapRegions])
	return r, nil
}

// AlignBatch aligns qs[i] against ts[i] for every i in one device call (new: a GPU needs batches).
// Sequences are copied into one flat blob: C never sees Go pointers inside Go memory (cgo pointer rules) and
// keeps nothing after the call returns.
func (algn *Aligner) AlignBatch(qs, ts [][]byte) ([]*AlignmentResult, []error) {
	n := len(qs)
	results := make([]*AlignmentResult, n)
	errs := make([]error, n)
	if n == 0 {
		return results, errs
	}
	qOff, tOff := make([]C.uint64_t, n), make([]C.uint64_t, n)
	qLen, tLen := make([]C.uint32_t, n), make([]C.uint32_t, n)
	total := 0
	for i := 0; i < n; i++ { // 16-byte aligned starts (lets the kernel stage with aligned dword loads)
		qOff[i] = C.uint64_t(total)
		qLen[i] = C.uint32_t(len(qs[i]))
		total += (len(qs[i]) + 15) &^ 15
		tOff[i] = C.uint64_t(total)
		tLen[i] = C.uint32_t(len(ts[i]))
		total += (len(ts[i]) + 15) &^ 15
	}
	blob := make([]byte, total+16)
	for i := 0; i < n; i++ {
		copy(blob[qOff[i]:], qs[i])
		copy(blob[tOff[i]:], ts[i])
	}
	p := algn.params()
	var out C.wfahip_results
	var rc C.int
	if algn.multi != nil { // a context set over several GPUs: contiguous shards, one host thread per GPU inside the library
		rc = C.wfahip_align_batch_multi(algn.multi, &p, (*C.uint8_t)(unsafe.Pointer(&blob[0])), C.uint64_t(len(blob)),
			&qOff[0], &qLen[0], &tOff[0], &tLen[0], C.uint64_t(n), &out)
	} else {
		rc = C.wfahip_align_batch(algn.ctx, &p, (*C.uint8_t)(unsafe.Pointer(&blob[0])), C.uint64_t(len(blob)),
			&qOff[0], &qLen[0], &tOff[0], &tLen[0], C.uint64_t(n), &out)
	}
	runtime.KeepAlive(blob)
	return algn.unpack(rc, &out, n)
}

// unpack copies a wfahip_results into pool-owned AlignmentResults and hands the C arrays back to the library.
func (algn *Aligner) unpack(rc C.int, out *C.wfahip_results, n int) ([]*AlignmentResult, []error) {
	results := make([]*AlignmentResult, n)
	errs := make([]error, n)
	if rc != 0 {
		err := fmt.Errorf("wfa: %s", C.GoString(C.wfahip_strerror(rc)))
		for i := range errs {
			errs[i] = err
		}
		return results, errs
	}
	defer C.wfahip_results_free(out)
	status := unsafe.Slice((*int32)(unsafe.Pointer(out.status)), n)
	score := unsafe.Slice((*uint32)(unsafe.Pointer(out.score)), n)
	tb := unsafe.Slice((*int32)(unsafe.Pointer(out.tbegin)), n)
	te := unsafe.Slice((*int32)(unsafe.Pointer(out.tend)), n)
	qb := unsafe.Slice((*int32)(unsafe.Pointer(out.qbegin)), n)
	qe := unsafe.Slice((*int32)(unsafe.Pointer(out.qend)), n)
	al := unsafe.Slice((*uint32)(unsafe.Pointer(out.align_len)), n)
	ma := unsafe.Slice((*uint32)(unsafe.Pointer(out.matches)), n)
	ga := unsafe.Slice((*uint32)(unsafe.Pointer(out.gaps)), n)
	gr := unsafe.Slice((*uint32)(unsafe.Pointer(out.gap_regions)), n)
	oo := unsafe.Slice((*uint64)(unsafe.Pointer(out.ops_off)), n)
	ol := unsafe.Slice((*uint32)(unsafe.Pointer(out.ops_len)), n)
	var ops []uint64
	if out.n_ops > 0 {
		ops = unsafe.Slice((*uint64)(unsafe.Pointer(out.ops)), int(out.n_ops))
	}
	for i := 0; i < n; i++ {
		switch uint32(status[i]) {
		case pairOK:
			r := NewAlignmentResult(algn.opt.GlobalAlignment)
			r.Ops = append(r.Ops[:0], ops[oo[i]:oo[i]+uint64(ol[i])]...) // copied into pool-owned memory
			r.Score = score[i]
			r.TBegin, r.TEnd, r.QBegin, r.QEnd = int(tb[i]), int(te[i]), int(qb[i]), int(qe[i])
			r.AlignLen, r.Matches, r.Gaps, r.GapRegions = al[i], ma[i], ga[i], gr[i]
			r.proccessed = true // ops arrive reversed + merged (process(), wfa_cigar.go:136-214, ran on the device)
			results[i] = r
		case pairEmpty:
			errs[i] = ErrEmptySeq
		case pairTooLong:
			errs[i] = ErrSeqTooLong
		default:
			errs[i] = fmt.Errorf("wfa: out of device memory for this pair")
		}
	}
	return results, errs
}

// Submit hands in one pair (the bytes are copied) and returns its ticket; Collect aligns everything submitted since
// the last Collect as ONE batch.  A per-pair loop like the reference's CLI (wfa-go/wfa-go.go:166-178) keeps its shape --
// Submit where it called Align, one Collect at the end -- and runs at batch throughput instead of one device round
// trip per pair.
//
// The error is non-nil when the pair could not be queued (context gone after RecycleAligner, out of host memory): the
// ticket is then meaningless and nothing was added, so the tickets of later pairs still line up with Collect.
func (algn *Aligner) Submit(q, t []byte) (uint64, error) {
	var ticket C.uint64_t
	var qp, tp *C.uint8_t
	if len(q) > 0 {
		qp = (*C.uint8_t)(unsafe.Pointer(&q[0]))
	}
	if len(t) > 0 {
		tp = (*C.uint8_t)(unsafe.Pointer(&t[0]))
	}
	rc := C.wfahip_submit(algn.ctx, qp, C.uint32_t(len(q)), tp, C.uint32_t(len(t)), &ticket)
	runtime.KeepAlive(q)
	runtime.KeepAlive(t)
	if rc != 0 {
		return 0, fmt.Errorf("wfa: %s", C.GoString(C.wfahip_strerror(rc)))
	}
	return uint64(ticket), nil
}

// Collect returns results[ticket], errs[ticket] for every pair submitted since the last Collect.
func (algn *Aligner) Collect() ([]*AlignmentResult, []error) {
	n := int(C.wfahip_pending(algn.ctx))
	p := algn.params()
	var out C.wfahip_results
	rc := C.wfahip_collect(algn.ctx, &p, &out)
	return algn.unpack(rc, &out, n)
}

// Wavefronts returns every stored M, I and D word (offset<<3 | type, wfa_wavefront.go:93) of one alignment, keyed by
// component, score and diagonal: what the reference keeps in Aligner.M/I/D (wfa.go:86) after Align.  A maintainer's
// Plot (wfa_component_plot.go:41) iterates exactly this; wfa_amd/aligner.py:plot_component is that function over the
// same data and reproduces the README tables from device wavefronts (tests/test_parity_gpu.py).
func (algn *Aligner) Wavefronts(q, t []byte) (map[byte]map[uint32]map[int]uint32, error) {
	var rows *C.wfahip_row
	var words *C.uint32_t
	var nRows, nWords C.uint64_t
	if len(q) == 0 || len(t) == 0 {
		return nil, ErrEmptySeq
	}
	p := algn.params()
	rc := C.wfahip_debug_wavefronts(algn.ctx, &p, (*C.uint8_t)(unsafe.Pointer(&q[0])), C.uint32_t(len(q)),
		(*C.uint8_t)(unsafe.Pointer(&t[0])), C.uint32_t(len(t)), &rows, &nRows, &words, &nWords, nil)
	if rc != 0 {
		return nil, fmt.Errorf("wfa: %s", C.GoString(C.wfahip_strerror(rc)))
	}
	defer C.wfahip_free(unsafe.Pointer(rows))
	defer C.wfahip_free(unsafe.Pointer(words))
	out := map[byte]map[uint32]map[int]uint32{'M': {}, 'I': {}, 'D': {}}
	rs := unsafe.Slice(rows, int(nRows))
	ws := unsafe.Slice((*uint32)(unsafe.Pointer(words)), int(nWords))
	for _, r := range rs {
		for ci, name := range []byte("MID") {
			for j := 0; j < int(r.width); j++ {
				if w := ws[int(r.word_off)+ci*int(r.width)+j]; w != 0 {
					if out[name][uint32(r.score)] == nil {
						out[name][uint32(r.score)] = map[int]uint32{}
					}
					out[name][uint32(r.score)][int(r.lo)+j] = w
				}
			}
		}
	}
	return out, nil
}

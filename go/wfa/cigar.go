package wfa

import (
	"strconv"
	"sync"
)

// AlignmentResult: same exported fields as the reference (wfa_cigar.go:30-48).
type AlignmentResult struct {
	Ops []uint64 // 24-bit null + 8 bit Op + 32-bit N

	Score uint32

	TBegin, TEnd int
	QBegin, QEnd int

	AlignLen   uint32
	Matches    uint32
	Gaps       uint32
	GapRegions uint32

	proccessed      bool
	globalAlignment bool
}

// Op extracts operation type and count.  (wfa_cigar.go:57-59)
func Op(op uint64) (byte, uint32) { return byte(op >> 32), uint32(op & MaskLower32) }

const OpM = uint64('M')
const OpD = uint64('D')
const OpI = uint64('I')
const OpX = uint64('X')
const OpH = uint64('H')
const MaskLower32 = 4294967295

var poolCIGAR = &sync.Pool{New: func() interface{} {
	return &AlignmentResult{Ops: make([]uint64, 0, 1024)}
}}

// NewAlignmentResult returns a result object from the pool.  (wfa_cigar.go:69-74)
func NewAlignmentResult(globalAlignment bool) *AlignmentResult {
	r := poolCIGAR.Get().(*AlignmentResult)
	r.Ops = r.Ops[:0]
	r.Score, r.AlignLen, r.Matches, r.Gaps, r.GapRegions = 0, 0, 0, 0, 0
	r.TBegin, r.TEnd, r.QBegin, r.QEnd = 0, 0, 0, 0
	r.proccessed = false
	r.globalAlignment = globalAlignment
	return r
}

// RecycleAlignmentResult recycles a result object.  (wfa_cigar.go:92-96)
func RecycleAlignmentResult(cigar *AlignmentResult) {
	if cigar != nil {
		poolCIGAR.Put(cigar)
	}
}

// alignedSpan returns the ops between the first and the last 'M' run (the reference's trimOps,
// wfa_cigar.go:217-233; like it, a list without any M yields an empty span here instead of a panic).
func alignedSpan(ops []uint64) []uint64 {
	first, last := len(ops), -1
	for i, op := range ops {
		if op>>32 != OpM {
			continue
		}
		if i < first {
			first = i
		}
		last = i
	}
	if last < 0 {
		return nil
	}
	return ops[first : last+1]
}

// CIGAR returns the CIGAR string; onlyAignedRegion drops the flanking clips/insertions.  (wfa_cigar.go:236-255)
func (cigar *AlignmentResult) CIGAR(onlyAignedRegion bool) string {
	ops := cigar.Ops
	if onlyAignedRegion {
		ops = alignedSpan(ops)
	}
	text := make([]byte, 0, 8*len(ops))
	for _, op := range ops {
		letter, count := Op(op)
		text = strconv.AppendUint(text, uint64(count), 10)
		text = append(text, letter)
	}
	return string(text)
}

var poolBytes = &sync.Pool{New: func() interface{} {
	buf := make([]byte, 0, 1024)
	return &buf
}}

// AlignmentText returns the formatted alignment text for Query, Alignment, and Target.  (wfa_cigar.go:259-333)
func (cigar *AlignmentResult) AlignmentText(q0, t0 *[]byte, onlyAignedRegion bool) (*[]byte, *[]byte, *[]byte) {
	var q, t []byte
	ops := cigar.Ops
	if !onlyAignedRegion {
		q, t = *q0, *t0
	} else {
		q = (*q0)[cigar.QBegin-1 : cigar.QEnd]
		t = (*t0)[cigar.TBegin-1 : cigar.TEnd]
		ops = alignedSpan(cigar.Ops)
	}
	Q, A, T := poolBytes.Get().(*[]byte), poolBytes.Get().(*[]byte), poolBytes.Get().(*[]byte)
	v, h := 0, 0
	for _, op := range ops {
		n := int(op & MaskLower32)
		switch op >> 32 {
		case OpM, OpX:
			bar := byte('|')
			if op>>32 == OpX {
				bar = ' '
			}
			for i := 0; i < n; i++ {
				*Q, *A, *T = append(*Q, q[v]), append(*A, bar), append(*T, t[h])
				v++
				h++
			}
		case OpI:
			for i := 0; i < n; i++ {
				*Q, *A, *T = append(*Q, '-'), append(*A, ' '), append(*T, t[h])
				h++
			}
		case OpD, OpH:
			for i := 0; i < n; i++ {
				*Q, *A, *T = append(*Q, q[v]), append(*A, ' '), append(*T, '-')
				v++
			}
		}
	}
	return Q, A, T
}

// RecycleAlignmentText recycles alignment text buffers.  (wfa_cigar.go:347-360)
func RecycleAlignmentText(Q, A, T *[]byte) {
	for _, b := range []*[]byte{Q, A, T} {
		if b != nil {
			*b = (*b)[:0]
			poolBytes.Put(b)
		}
	}
}

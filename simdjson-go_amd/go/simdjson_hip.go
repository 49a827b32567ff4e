//go:build hip && cgo
// +build hip,cgo

// simdjson_hip.go -- third backend of the reference's Go package: the MI355X engine (libsjhip).
//
// Drop this file into the root of github.com/minio/simdjson-go and build with
//     CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/simdjson-go_amd -lsjhip" go build -tags hip
// The two existing backend files must exclude the tag (INTEGRATION.md section 2 lists the complete edits):
//     simdjson_amd64.go   //go:build !appengine && !noasm && gc && !hip
//     simdjson_other.go   //go:build (!amd64 || appengine || !gc || noasm) && !hip
// No other reference file changes: this file declares no identifier that an untagged reference file (or a
// *_amd64.go file, which stays in the build on amd64) declares -- tools/check_go_collisions.py checks that against
// the identifier inventory of the reference (tests/golden/go_reference_symbols.json, tests/test_go_binding.py).
//
// It provides exactly the backend symbol set of simdjson_other.go:29-76 (SupportedCPU, Parse, ParseND, Stream,
// ParseNDStream) plus newInternalParsedJson (simdjson_amd64.go:41, which the excluded file no longer provides).
// It uses the reference's own types: ParsedJson / TStrings (parsed_json.go:60-71), internalParsedJson
// (parsed_json.go:83-93: the embedded ParsedJson and copyStrings, which WithCopyStrings sets, options.go:13-18) and
// ParserOption (options.go:4).  Everything above the tape -- Iter, Object, Array, the Serializer -- is unchanged Go
// code that only reads ParsedJson{Message, Tape, Strings}.
//
// NOTE: there is no Go toolchain in the build container of this repository, so this file has
// not been compiled there; the Python mirror (simdjson-go_amd/sjhip) binds the same C symbols
// through ctypes and tests/test_cabi_caller.py drives this file's exact call sequences from a C program.

package simdjson

/*
#include <stdlib.h>
#include "sjhip.h"
*/
import "C"

import (
	"bufio"
	"errors"
	"fmt"
	"io"
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"
)

// SupportedCPU reports whether the backend can run: a gfx950 device is visible.
// (Name kept for API compatibility, simdjson_amd64.go:37.)
func SupportedCPU() bool {
	return C.sjhip_supported() != 0
}

// hipCtx wraps one sjhip_ctx (HIP stream + recycled device arenas).  A context serves one
// parse at a time, like one reference `internalParsedJson`.
type hipCtx struct {
	h *C.sjhip_ctx
}

// New contexts are created round robin over the visible devices: concurrent Parse calls (and the blocks of a
// stream) spread over every GPU of the node.
var nextDevice uint32

var ctxPool = sync.Pool{New: func() interface{} {
	n := int(C.sjhip_device_count())
	if n < 1 {
		return (*hipCtx)(nil)
	}
	dev := int(atomic.AddUint32(&nextDevice, 1)-1) % n
	h := C.sjhip_ctx_create(C.int(dev))
	if h == nil {
		return (*hipCtx)(nil)
	}
	c := &hipCtx{h: h}
	runtime.SetFinalizer(c, func(c *hipCtx) { C.sjhip_ctx_destroy(c.h) })
	return c
}}

// newInternalParsedJson is simdjson_amd64.go:41-62 for this backend: the reference's own internalParsedJson
// (parsed_json.go:83-93) carries the options and is recycled through reuse.internal exactly as there.
func newInternalParsedJson(reuse *ParsedJson, opts []ParserOption) (*internalParsedJson, error) {
	if !SupportedCPU() {
		return nil, errors.New("Host CPU does not meet target specs")
	}
	var pj *internalParsedJson
	if reuse != nil && reuse.internal != nil {
		pj = reuse.internal
		pj.ParsedJson = *reuse // recycle Tape / Strings capacity (simdjson_amd64.go:46-51)
		pj.ParsedJson.internal = nil
	} else {
		pj = &internalParsedJson{}
		if reuse != nil {
			pj.ParsedJson = *reuse
			pj.ParsedJson.internal = nil
		}
	}
	pj.copyStrings = true
	for _, opt := range opts {
		if err := opt(pj); err != nil {
			return nil, err
		}
	}
	return pj, nil
}

// parseMessageHip is (*internalParsedJson).parseMessage (parse_json_amd64.go:52) on the GPU.  (Its own name: the
// reference's method stays in the build on amd64.)
func (pj *internalParsedJson) parseMessageHip(msg []byte, ndjson bool) error {
	c, _ := ctxPool.Get().(*hipCtx)
	if c == nil {
		return errors.New("Host CPU does not meet target specs")
	}
	defer ctxPool.Put(c)

	var flags C.uint32_t
	if ndjson {
		flags |= C.SJHIP_FLAG_NDJSON
	}
	if pj.copyStrings {
		flags |= C.SJHIP_FLAG_COPY_STRINGS
	}
	var tapeLen, stringsLen, msgOff, msgLen C.size_t
	var p *C.uint8_t
	if len(msg) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&msg[0]))
	}
	rc := C.sjhip_parse(c.h, p, C.size_t(len(msg)), flags, &tapeLen, &stringsLen, &msgOff, &msgLen)
	runtime.KeepAlive(msg)
	switch rc {
	case C.SJHIP_OK:
	case C.SJHIP_ERR_STAGE1:
		return errors.New("Failed to find all structural indices for stage 1")
	case C.SJHIP_ERR_STAGE2:
		return errors.New("Bad parsing while executing stage 2")
	case C.SJHIP_ERR_NODEVICE:
		return errors.New("Host CPU does not meet target specs")
	default:
		return fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(c.h)))
	}

	// pj.Message aliases the caller's buffer exactly like bytes.TrimSpace does (:55)
	pj.Message = msg[int(msgOff) : int(msgOff)+int(msgLen)]
	if cap(pj.Tape) < int(tapeLen) {
		pj.Tape = make([]uint64, int(tapeLen))
	}
	pj.Tape = pj.Tape[:int(tapeLen)]
	if pj.Strings == nil {
		pj.Strings = &TStrings{}
	}
	if cap(pj.Strings.B) < int(stringsLen) {
		pj.Strings.B = make([]byte, int(stringsLen))
	}
	pj.Strings.B = pj.Strings.B[:int(stringsLen)]
	var tp *C.uint64_t
	var sp *C.uint8_t
	if tapeLen > 0 {
		tp = (*C.uint64_t)(unsafe.Pointer(&pj.Tape[0]))
	}
	if stringsLen > 0 {
		sp = (*C.uint8_t)(unsafe.Pointer(&pj.Strings.B[0]))
	}
	if rc := C.sjhip_fetch(c.h, tp, sp); rc != C.SJHIP_OK {
		return fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(c.h)))
	}
	// a pooled context keeps arenas sized for the largest message it has seen: give them back after an unusual one
	if uint64(C.sjhip_ctx_device_bytes(c.h)) > PoolTrimBytes {
		C.sjhip_ctx_trim(c.h)
	}
	return nil
}

// PoolTrimBytes: device memory a pooled context may keep between parses (its arenas are ~13-20x the largest message
// it has parsed, INTEGRATION.md section 3c); beyond it the context is trimmed before it goes back to the pool.
var PoolTrimBytes uint64 = 16 << 30

// ParseBatch parses many documents with one launch set (sjhip_parse_batch): the returned ParsedJson holds document i
// as root i -- iterate with pj.Iter() / Advance() as over a ParseND result.  It replaces the goroutine-per-Parse shape
// of benchmarks_test.go:60-75 where the documents are small: a GPU parse has a fixed cost per call.  One invalid
// document fails the batch with that document's error.  Strings are always copied (Message stays empty).
func ParseBatch(docs [][]byte, reuse *ParsedJson) (*ParsedJson, error) {
	pj, err := newInternalParsedJson(reuse, nil)
	if err != nil {
		return nil, err
	}
	c, _ := ctxPool.Get().(*hipCtx)
	if c == nil {
		return nil, errors.New("Host CPU does not meet target specs")
	}
	defer ctxPool.Put(c)
	n := len(docs)
	// the pointer array lives in C memory: cgo forbids passing a Go slice of Go pointers; the documents are pinned
	// for the duration of the call (the library copies them to the device before it returns)
	var pinner runtime.Pinner
	defer pinner.Unpin()
	ptrs := (*[1 << 28]*C.uint8_t)(C.malloc(C.size_t(n+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	lens := (*[1 << 28]C.size_t)(C.malloc(C.size_t(n+1) * C.size_t(unsafe.Sizeof(C.size_t(0)))))
	defer C.free(unsafe.Pointer(ptrs))
	defer C.free(unsafe.Pointer(lens))
	for i, d := range docs {
		if len(d) > 0 {
			pinner.Pin(&d[0])
			ptrs[i] = (*C.uint8_t)(unsafe.Pointer(&d[0]))
		} else {
			ptrs[i] = nil
		}
		lens[i] = C.size_t(len(d))
	}
	var tapeLen, stringsLen C.size_t
	rc := C.sjhip_parse_batch(c.h, (**C.uint8_t)(unsafe.Pointer(ptrs)), (*C.size_t)(unsafe.Pointer(lens)), C.size_t(n),
		C.SJHIP_FLAG_COPY_STRINGS, &tapeLen, &stringsLen)
	runtime.KeepAlive(docs)
	switch rc {
	case C.SJHIP_OK:
	case C.SJHIP_ERR_STAGE1:
		return nil, errors.New("Failed to find all structural indices for stage 1")
	case C.SJHIP_ERR_STAGE2:
		return nil, errors.New("Bad parsing while executing stage 2")
	default:
		return nil, fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(c.h)))
	}
	pj.Message = pj.Message[:0]
	if cap(pj.Tape) < int(tapeLen) {
		pj.Tape = make([]uint64, int(tapeLen))
	}
	pj.Tape = pj.Tape[:int(tapeLen)]
	if pj.Strings == nil {
		pj.Strings = &TStrings{}
	}
	if cap(pj.Strings.B) < int(stringsLen) {
		pj.Strings.B = make([]byte, int(stringsLen))
	}
	pj.Strings.B = pj.Strings.B[:int(stringsLen)]
	var tp *C.uint64_t
	var sp *C.uint8_t
	if tapeLen > 0 {
		tp = (*C.uint64_t)(unsafe.Pointer(&pj.Tape[0]))
	}
	if stringsLen > 0 {
		sp = (*C.uint8_t)(unsafe.Pointer(&pj.Strings.B[0]))
	}
	if rc := C.sjhip_fetch(c.h, tp, sp); rc != C.SJHIP_OK {
		return nil, fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(c.h)))
	}
	return &pj.ParsedJson, nil
}

// Parse an object or array from a block of data and return the parsed JSON (simdjson_amd64.go:66).
func Parse(b []byte, reuse *ParsedJson, opts ...ParserOption) (*ParsedJson, error) {
	pj, err := newInternalParsedJson(reuse, opts)
	if err != nil {
		return nil, err
	}
	if err = pj.parseMessageHip(b, false); err != nil {
		return nil, err
	}
	parsed := &pj.ParsedJson
	parsed.internal = pj
	return parsed, nil
}

// ViewParser is Parse / ParseND for the recycling pattern (`reuse`, simdjson_amd64.go:46-51) without the copy of the
// result into Go memory: it owns one GPU context, and the ParsedJson it returns has Tape and Strings.B aliasing that
// context's pinned host memory (sjhip_fetch_view) -- overwritten by the next call on the same ViewParser, exactly as a
// recycled ParsedJson is.  The slices must not be appended to or written, and the ParsedJson must not be handed to
// Parse / ParseND as `reuse` (they would recycle slices that are not Go memory).  One ViewParser per goroutine; keep it
// reachable for as long as slices of its ParsedJson are in use (the context, and with it the pinned memory, is released
// when the ViewParser is collected -- holding the *ParsedJson it returned is enough: it points into the ViewParser).
// Parse(twitter.json) host to host: 152 us through Parse, ~120 us through a ViewParser (DESIGN.md section 5).
type ViewParser struct {
	c  *hipCtx
	pj internalParsedJson
}

// NewViewParser binds a context (round robin over the visible GPUs, like the pool).
func NewViewParser(opts ...ParserOption) (*ViewParser, error) {
	if !SupportedCPU() {
		return nil, errors.New("Host CPU does not meet target specs")
	}
	c, _ := ctxPool.New().(*hipCtx)
	if c == nil {
		return nil, errors.New("Host CPU does not meet target specs")
	}
	v := &ViewParser{c: c}
	v.pj.copyStrings = true
	for _, opt := range opts {
		if err := opt(&v.pj); err != nil {
			return nil, err
		}
	}
	return v, nil
}

func (v *ViewParser) parse(msg []byte, ndjson bool) (*ParsedJson, error) {
	var flags C.uint32_t
	if ndjson {
		flags |= C.SJHIP_FLAG_NDJSON
	}
	if v.pj.copyStrings {
		flags |= C.SJHIP_FLAG_COPY_STRINGS
	}
	var tapeLen, stringsLen, msgOff, msgLen C.size_t
	var p *C.uint8_t
	if len(msg) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&msg[0]))
	}
	rc := C.sjhip_parse(v.c.h, p, C.size_t(len(msg)), flags, &tapeLen, &stringsLen, &msgOff, &msgLen)
	runtime.KeepAlive(msg)
	switch rc {
	case C.SJHIP_OK:
	case C.SJHIP_ERR_STAGE1:
		return nil, errors.New("Failed to find all structural indices for stage 1")
	case C.SJHIP_ERR_STAGE2:
		return nil, errors.New("Bad parsing while executing stage 2")
	default:
		return nil, fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(v.c.h)))
	}
	var tp *C.uint64_t
	var sp *C.uint8_t
	if rc := C.sjhip_fetch_view(v.c.h, &tp, &sp); rc != C.SJHIP_OK {
		return nil, fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_last_error(v.c.h)))
	}
	pj := &v.pj.ParsedJson
	pj.Message = msg[int(msgOff) : int(msgOff)+int(msgLen)]
	pj.Tape = nil
	if tapeLen > 0 {
		pj.Tape = unsafe.Slice((*uint64)(unsafe.Pointer(tp)), int(tapeLen))
	}
	if pj.Strings == nil {
		pj.Strings = &TStrings{}
	}
	pj.Strings.B = nil
	if stringsLen > 0 {
		pj.Strings.B = unsafe.Slice((*byte)(unsafe.Pointer(sp)), int(stringsLen))
	}
	return pj, nil
}

// InputBlock returns n bytes of pinned host memory of the parser's context to read the next message into (a file, a
// socket) and hand to Parse / ParseND: the copy to the device then runs at the pinned rate (sjhip_input_block).  Valid
// until the next InputBlock call with a larger n.
func (v *ViewParser) InputBlock(n int) []byte {
	p := C.sjhip_input_block(v.c.h, C.size_t(n))
	if p == nil {
		return nil
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(p)), n)
}

// Parse is Parse(b, reuse) with the ViewParser's own ParsedJson as `reuse`.
func (v *ViewParser) Parse(b []byte) (*ParsedJson, error) { return v.parse(b, false) }

// ParseND is ParseND(b, reuse) with the ViewParser's own ParsedJson as `reuse`.
func (v *ViewParser) ParseND(b []byte) (*ParsedJson, error) { return v.parse(b, true) }

// multiPool holds handles that own one context per visible GPU (sjhip_multi_*): ParseND of a large message is cut at
// record boundaries into one shard per device inside the library.
var multiPool = sync.Pool{New: func() interface{} {
	h := C.sjhip_multi_create(nil, 0)
	if h == nil {
		return (*hipMulti)(nil)
	}
	m := &hipMulti{h: h}
	runtime.SetFinalizer(m, func(m *hipMulti) { C.sjhip_multi_destroy(m.h) })
	return m
}}

type hipMulti struct{ h *C.sjhip_multi }

// messages below this size stay on one device (a shard should keep a GPU busy for longer than its fixed costs)
const multiMinBytes = 32 << 20

// parseMessageMulti is parseMessageHip(msg, true) over every GPU of the node.
func (pj *internalParsedJson) parseMessageMulti(msg []byte) error {
	m, _ := multiPool.Get().(*hipMulti)
	if m == nil {
		return errors.New("Host CPU does not meet target specs")
	}
	defer multiPool.Put(m)
	var flags C.uint32_t = C.SJHIP_FLAG_NDJSON
	if pj.copyStrings {
		flags |= C.SJHIP_FLAG_COPY_STRINGS
	}
	var tapeLen, stringsLen, msgOff, msgLen C.size_t
	rc := C.sjhip_parse_nd_multi(m.h, (*C.uint8_t)(unsafe.Pointer(&msg[0])), C.size_t(len(msg)), flags, &tapeLen, &stringsLen, &msgOff, &msgLen)
	runtime.KeepAlive(msg)
	switch rc {
	case C.SJHIP_OK:
	case C.SJHIP_ERR_STAGE1:
		return errors.New("Failed to find all structural indices for stage 1")
	case C.SJHIP_ERR_STAGE2:
		return errors.New("Bad parsing while executing stage 2")
	default:
		return fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_multi_last_error(m.h)))
	}
	pj.Message = msg[int(msgOff) : int(msgOff)+int(msgLen)]
	if cap(pj.Tape) < int(tapeLen) {
		pj.Tape = make([]uint64, int(tapeLen))
	}
	pj.Tape = pj.Tape[:int(tapeLen)]
	if pj.Strings == nil {
		pj.Strings = &TStrings{}
	}
	if cap(pj.Strings.B) < int(stringsLen) {
		pj.Strings.B = make([]byte, int(stringsLen))
	}
	pj.Strings.B = pj.Strings.B[:int(stringsLen)]
	var tp *C.uint64_t
	var sp *C.uint8_t
	if tapeLen > 0 {
		tp = (*C.uint64_t)(unsafe.Pointer(&pj.Tape[0]))
	}
	if stringsLen > 0 {
		sp = (*C.uint8_t)(unsafe.Pointer(&pj.Strings.B[0]))
	}
	if rc := C.sjhip_fetch_multi(m.h, tp, sp); rc != C.SJHIP_OK {
		return fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_multi_last_error(m.h)))
	}
	return nil
}

// ParseND will parse newline delimited JSON objects or arrays (simdjson_amd64.go:82).  On a node with several GPUs a
// large message is sharded over all of them at record boundaries (sjhip_parse_nd_multi); the result is the same
// ParsedJson bit for bit.
func ParseND(b []byte, reuse *ParsedJson, opts ...ParserOption) (*ParsedJson, error) {
	pj, err := newInternalParsedJson(reuse, opts)
	if err != nil {
		return nil, err
	}
	if len(b) >= multiMinBytes && int(C.sjhip_device_count()) > 1 {
		err = pj.parseMessageMulti(b)
	} else {
		err = pj.parseMessageHip(b, true)
	}
	if err != nil {
		return nil, err
	}
	return &pj.ParsedJson, nil
}

// A Stream is used to stream back results (simdjson_amd64.go:96-99).
type Stream struct {
	Value *ParsedJson
	Error error
}

// ParseNDStream keeps the contract of simdjson_amd64.go:116-216: the input is cut into blocks of about 10 MiB that
// end on a record boundary, every block is parsed as an independent ND document with every string copied, results
// arrive on res in input order, and the first error (io.EOF at the end of the input) is the last value sent before
// res is closed.  The pipeline itself lives in the library (sjhip_stream_*, csrc/stream_api.hip): the reader fills
// pinned blocks directly, every block in flight has its own context and HIP stream on one of the node's GPUs (round
// robin), and results are copied out of pinned memory into the (recycled) ParsedJson.
func ParseNDStream(r io.Reader, res chan<- Stream, reuse <-chan *ParsedJson) {
	parseNDStreamHip(r, res, reuse, false)
}

// ParseNDStreamInPlace is ParseNDStream without the copy of every result out of the stream's pinned memory (2.4 bytes per
// input byte on parking-citations: 13 GB/s with four copying threads, 19 GB/s without the copy, DESIGN.md section 5a):
// Stream.Value's Tape, Strings.B and Message alias memory of the stream until the consumer hands the value back on
// `reuse` -- which it MUST do for every value, before the next one is delivered (the blocks behind it keep being read
// and parsed meanwhile).  `reuse` must not be nil.  The value handed back is only a token: nothing of it is recycled.
// CLOSING `reuse` ends the stream and INVALIDATES every value it delivered (their slices alias pinned memory that is freed
// with the stream): copy what must outlive the stream before closing.
func ParseNDStreamInPlace(r io.Reader, res chan<- Stream, reuse <-chan *ParsedJson) {
	if reuse == nil {
		go func() {
			res <- Stream{Error: errors.New("ParseNDStreamInPlace: the reuse channel is how blocks are released, it must not be nil")}
			close(res)
		}()
		return
	}
	parseNDStreamHip(r, res, reuse, true)
}

func parseNDStreamHip(r io.Reader, res chan<- Stream, reuse <-chan *ParsedJson, inPlace bool) {
	if !SupportedCPU() {
		go func() {
			res <- Stream{Error: errors.New("Host CPU does not meet target specs")}
			close(res)
		}()
		return
	}
	const blockSize = 10 << 20 // tmpSize, simdjson_amd64.go:127
	const reserve = blockSize/8 + 64<<10
	st := C.sjhip_stream_create(0, 0, C.size_t(blockSize+reserve), 0, 0)
	if st == nil {
		go func() {
			res <- Stream{Error: errors.New("Host CPU does not meet target specs")}
			close(res)
		}()
		return
	}
	// Feeding and delivery are decoupled like in the reference (a reader goroutine queues blocks, results are sent as
	// soon as their block is parsed, simdjson_amd64.go:131-216): the library allows one submitting and one taking
	// thread.  `submitted` wakes the deliverer when a block was queued or the reader has finished, `freed` wakes the
	// reader when a slot was released.
	submitted := make(chan struct{}, 1)
	freed := make(chan struct{}, 1)
	stop := make(chan struct{}) // closed by the deliverer when the stream ends with an error
	readDone := make(chan error, 1)
	readerExited := make(chan struct{}) // the stream is destroyed only once the reader no longer touches it
	notify := func(c chan struct{}) {
		select {
		case c <- struct{}{}:
		default:
		}
	}

	// reader: fills pinned blocks and submits them
	go func() {
		defer close(readerExited)
		rd := bufio.NewReaderSize(r, blockSize)
		for {
			var blk *C.uint8_t
			var capacity C.size_t
			rc := C.sjhip_stream_acquire(st, &blk, &capacity)
			if rc == C.SJHIP_STREAM_FULL { // every slot holds a block: wait for the deliverer to release one
				select {
				case <-freed:
					continue
				case <-stop:
					return
				}
			}
			if rc != C.SJHIP_OK {
				if rc == C.SJHIP_ERR_STREAM_CLOSED { // a block has failed: the deliverer reports that block's error
					readDone <- nil
				} else { // an internal error: it must reach the consumer, or res would be closed without a final value
					readDone <- fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_stream_last_error(st)))
				}
				notify(submitted)
				return
			}
			buf := unsafe.Slice((*byte)(unsafe.Pointer(blk)), int(capacity))
			n, rerr := io.ReadFull(rd, buf[:blockSize]) // straight into pinned memory (tmpPool's role)
			if rerr == nil {                            // a full block: extend it to the end of the current record
				rest, lerr := rd.ReadBytes('\n')
				if n+len(rest) > int(capacity) { // a record longer than the reserve: a larger pinned block
					if C.sjhip_stream_grow(st, C.size_t(n), C.size_t(n+len(rest)), &blk) != C.SJHIP_OK {
						C.sjhip_stream_cancel(st)
						readDone <- fmt.Errorf("sjhip: %s", C.GoString(C.sjhip_stream_last_error(st)))
						notify(submitted)
						return
					}
					buf = unsafe.Slice((*byte)(unsafe.Pointer(blk)), n+len(rest))
				}
				copy(buf[n:], rest)
				n += len(rest)
				if lerr != nil {
					rerr = lerr
				}
			} else if rerr == io.ErrUnexpectedEOF {
				rerr = io.EOF
			}
			if n > 0 { // `if len(tmp) > 0`, :178
				C.sjhip_stream_submit(st, C.size_t(n))
			} else {
				C.sjhip_stream_cancel(st)
			}
			if rerr != nil {
				readDone <- rerr // io.EOF on a clean end
				notify(submitted)
				return
			}
			notify(submitted)
		}
	}()

	// deliverer: takes results in submission order and sends them at once
	go func() {
		defer close(res)
		defer func() {
			go func() {
				<-readerExited
				C.sjhip_stream_destroy(st)
			}()
		}()
		var finalErr error
		finished := false
		for {
			var out C.sjhip_stream_result
			rc := C.sjhip_stream_next(st, &out) // waits for the oldest outstanding block
			switch rc {
			case C.SJHIP_OK:
			case C.SJHIP_STREAM_EMPTY:
				if finished {
					if finalErr != nil {
						res <- Stream{Error: finalErr}
					}
					return
				}
				select { // nothing outstanding: wait for the reader
				case <-submitted:
				case finalErr = <-readDone:
					finished = true
				}
				continue
			case C.SJHIP_ERR_STAGE1:
				close(stop)
				res <- Stream{Error: fmt.Errorf("parsing input: %w", errors.New("Failed to find all structural indices for stage 1"))}
				return
			case C.SJHIP_ERR_STAGE2:
				close(stop)
				res <- Stream{Error: fmt.Errorf("parsing input: %w", errors.New("Bad parsing while executing stage 2"))}
				return
			default:
				close(stop)
				res <- Stream{Error: fmt.Errorf("parsing input: sjhip: %s", C.GoString(C.sjhip_stream_last_error(st)))}
				return
			}
			if inPlace { // the consumer reads the block where the DMA left it and says when it is done
				tl, sl, ml := int(out.tape_len), int(out.strings_len), int(out.message_len)
				v := &ParsedJson{Strings: &TStrings{}}
				if tl > 0 {
					v.Tape = unsafe.Slice((*uint64)(unsafe.Pointer(out.tape)), tl)
				}
				if sl > 0 {
					v.Strings.B = unsafe.Slice((*byte)(unsafe.Pointer(out.strings)), sl)
				}
				if ml > 0 {
					v.Message = unsafe.Slice((*byte)(unsafe.Pointer(out.message)), ml)
				}
				res <- Stream{Value: v}
				if _, ok := <-reuse; !ok {
					// the consumer closed `reuse` (legal with ParseNDStream): that is NOT a hand-back.  In-place mode has no
					// copy to fall back on: the value delivered last aliases pinned memory of the stream, which is torn down
					// when this goroutine returns -- CLOSING `reuse` THEREFORE INVALIDATES EVERY VALUE THIS STREAM DELIVERED
					// (documented on ParseNDStreamInPlace; a consumer that wants to keep a value copies it before closing).
					// The error is offered without blocking: a consumer that has stopped reading `res` must not pin this
					// goroutine and the stream forever.
					close(stop)
					select {
					case res <- Stream{Error: errors.New("ParseNDStreamInPlace: reuse channel closed; values delivered by this stream are no longer valid")}:
					default:
					}
					return
				}
				C.sjhip_stream_release(st)
				notify(freed)
				continue
			}
			var pj ParsedJson
			select { // `select { case v := <-reuse: ... default: }`, simdjson_amd64.go:181-190
			case old := <-reuse:
				if old != nil {
					pj = *old
				}
			default:
			}
			tl, sl, ml := int(out.tape_len), int(out.strings_len), int(out.message_len)
			if cap(pj.Tape) < tl {
				pj.Tape = make([]uint64, tl)
			}
			pj.Tape = pj.Tape[:tl]
			if pj.Strings == nil {
				pj.Strings = &TStrings{}
			}
			if cap(pj.Strings.B) < sl {
				pj.Strings.B = make([]byte, sl)
			}
			pj.Strings.B = pj.Strings.B[:sl]
			if cap(pj.Message) < ml {
				pj.Message = make([]byte, ml)
			}
			pj.Message = pj.Message[:ml]
			// the three copies out of pinned memory side by side: one thread moves 5-6 GB/s, and a block's result is
			// 2.4 bytes per input byte (four copying threads: 13 GB/s end to end, tools/stream_bench.py)
			var wg sync.WaitGroup
			if tl > 0 {
				wg.Add(1)
				go func() {
					defer wg.Done()
					copy(pj.Tape, unsafe.Slice((*uint64)(unsafe.Pointer(out.tape)), tl))
				}()
			}
			if sl > 0 {
				wg.Add(1)
				go func() {
					defer wg.Done()
					copy(pj.Strings.B, unsafe.Slice((*byte)(unsafe.Pointer(out.strings)), sl))
				}()
			}
			if ml > 0 {
				copy(pj.Message, unsafe.Slice((*byte)(unsafe.Pointer(out.message)), ml))
			}
			wg.Wait()
			C.sjhip_stream_release(st)
			notify(freed)
			res <- Stream{Value: &pj}
		}
	}()
}

//go:build hip && cgo
// +build hip,cgo

// simdjson_hip.go -- third backend of package simdjson: the MI355X engine (libsjhip).
//
// Drop this file into the root of github.com/minio/simdjson-go and build with
//     CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/simdjson-go_amd -lsjhip" go build -tags hip
// The two existing backend files must exclude the tag (see INTEGRATION.md):
//     simdjson_amd64.go   //go:build !appengine && !noasm && gc && !hip
//     simdjson_other.go   //go:build (!amd64 || appengine || !gc || noasm) && !hip
//
// It provides exactly the backend symbol set of simdjson_other.go:29-76 (SupportedCPU, Parse,
// ParseND, Stream, ParseNDStream).  Everything above the tape -- Iter, Object, Array, the
// Serializer -- is unchanged Go code that only reads ParsedJson{Message, Tape, Strings}.
//
// NOTE: there is no Go toolchain in the build container of this repository, so this file has
// not been compiled there; the Python mirror (simdjson-go_amd/sjhip) binds the same C symbols
// through ctypes and is what the test-suite executes.

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

var ctxPool = sync.Pool{New: func() interface{} {
	h := C.sjhip_ctx_create(0)
	if h == nil {
		return (*hipCtx)(nil)
	}
	c := &hipCtx{h: h}
	runtime.SetFinalizer(c, func(c *hipCtx) { C.sjhip_ctx_destroy(c.h) })
	return c
}}

type internalParsedJson struct {
	ParsedJson
	copyStrings bool
}

func newInternalParsedJson(reuse *ParsedJson, opts []ParserOption) (*internalParsedJson, error) {
	if !SupportedCPU() {
		return nil, errors.New("Host CPU does not meet target specs")
	}
	pj := &internalParsedJson{}
	if reuse != nil {
		pj.ParsedJson = *reuse // recycle Tape / Strings capacity (simdjson_amd64.go:46-51)
		pj.ParsedJson.internal = nil
	}
	pj.copyStrings = true
	for _, opt := range opts {
		if err := opt(pj); err != nil {
			return nil, err
		}
	}
	return pj, nil
}

// parseMessage mirrors (*internalParsedJson).parseMessage (parse_json_amd64.go:52).
func (pj *internalParsedJson) parseMessage(msg []byte, ndjson bool) error {
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
	return nil
}

// Parse an object or array from a block of data and return the parsed JSON (simdjson_amd64.go:66).
func Parse(b []byte, reuse *ParsedJson, opts ...ParserOption) (*ParsedJson, error) {
	pj, err := newInternalParsedJson(reuse, opts)
	if err != nil {
		return nil, err
	}
	if err = pj.parseMessage(b, false); err != nil {
		return nil, err
	}
	parsed := &pj.ParsedJson
	parsed.internal = pj
	return parsed, nil
}

// ParseND will parse newline delimited JSON objects or arrays (simdjson_amd64.go:82).
func ParseND(b []byte, reuse *ParsedJson, opts ...ParserOption) (*ParsedJson, error) {
	pj, err := newInternalParsedJson(reuse, opts)
	if err != nil {
		return nil, err
	}
	if err = pj.parseMessage(b, true); err != nil {
		return nil, err
	}
	return &pj.ParsedJson, nil
}

// A Stream is used to stream back results (simdjson_amd64.go:96-99).
type Stream struct {
	Value *ParsedJson
	Error error
}

// ParseNDStream keeps the contract of simdjson_amd64.go:116-216: the input is cut into blocks of
// about 10 MiB that end on a record boundary, every block is parsed as an independent ND
// document, results arrive on res in input order, and the first error (io.EOF at the end of the
// input) is the last value sent before res is closed.  Blocks are independent, so with several
// GPUs the context pool can hand out contexts living on different devices.
func ParseNDStream(r io.Reader, res chan<- Stream, reuse <-chan *ParsedJson) {
	if !SupportedCPU() {
		go func() {
			res <- Stream{Error: errors.New("Host CPU does not meet target specs")}
			close(res)
		}()
		return
	}
	const blockSize = 10 << 20
	inFlight := (runtime.GOMAXPROCS(0) + 1) / 2
	if inFlight < 1 {
		inFlight = 1
	}
	// ordered: one single-slot mailbox per block, consumed in submission order
	ordered := make(chan chan Stream, inFlight)

	go func() { // deliverer
		defer close(res)
		failed := false
		for box := range ordered {
			out := <-box
			if !failed {
				res <- out
			}
			if out.Error != nil {
				failed = true
			}
		}
	}()

	go func() { // block cutter + dispatcher
		defer close(ordered)
		rd := bufio.NewReaderSize(r, blockSize)
		submit := func(s Stream) {
			box := make(chan Stream, 1)
			box <- s
			ordered <- box
		}
		for {
			block := make([]byte, blockSize, blockSize+4096)
			n, rerr := io.ReadFull(rd, block)
			block = block[:n]
			if rerr == nil { // a full block: extend it to the end of the current record
				rest, lerr := rd.ReadBytes('\n')
				block = append(block, rest...)
				if lerr != nil {
					rerr = lerr
				}
			} else if rerr == io.ErrUnexpectedEOF {
				rerr = io.EOF
			}
			if len(block) > 0 {
				box := make(chan Stream, 1)
				ordered <- box
				go func(data []byte) {
					pj := internalParsedJson{copyStrings: true}
					select {
					case old := <-reuse:
						if old != nil {
							pj.ParsedJson = *old
						}
					default:
					}
					if err := pj.parseMessage(data, true); err != nil {
						box <- Stream{Error: fmt.Errorf("parsing input: %w", err)}
						return
					}
					done := pj.ParsedJson
					box <- Stream{Value: &done}
				}(block)
			}
			if rerr != nil {
				submit(Stream{Error: rerr}) // io.EOF on a clean end
				return
			}
		}
	}()
}

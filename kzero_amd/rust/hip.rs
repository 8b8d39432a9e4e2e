//! kz-core/src/network/hip.rs — `HipNetwork`: the MI355X executor behind kZero's `Network` trait.
//!
//! Drop-in sibling of `CudaNetwork` (kz-core/src/network/cudnn.rs:18-88): same constructor shape, same
//! `Network<B>` contract (kz-core/src/network/mod.rs:52-63), same `decode_output` (network/common.rs:16-100).
//! The arithmetic lives in libkzhip.so (C ABI: include/kz_hip.h); this file only binds it.
//!
//! Differences from `CudaNetwork`, all invisible to callers:
//!  * the mapper's *packed* output (`InputMapper::encode_input`: BitBuffer + scalars, mapping/mod.rs:37) is handed
//!    over as is; the dense f32 expansion (`encode_input_full`, mapping/mod.rs:40-63) happens on the GPU;
//!  * no NaN padding (cudnn.rs:65) up to `max_batch_size` and no input clone (cudnn.rs:70): only `batch` rows exist;
//!  * `decode_output`'s gather + softmax (common.rs:60-86) runs on the GPU by default (`KZ_HIP_DECODE=device`): the
//!    executor thread builds the `move_to_index` list of every board when it submits the batch, the GPU returns
//!    tanh(value), softmax(wdl) and the per-move probabilities — 0.2 KB instead of 7.5 KB per chess evaluation over
//!    PCIe and no softmax on this thread.  Measured with the C++ mirror of this file (tests/cpp/bench_executor.cpp,
//!    chess 20x256 f16, ONE executor thread, move generation + a SipHash lookup per move on it): host decode
//!    257-350k evals/s box to box (the thread is the bottleneck: 3.2-3.9 CPU-s per million evaluations), device decode
//!    404-505k with the thread working 0.8-0.9 of a core, the GPU alone 482-527k.  `KZ_HIP_DECODE=host` keeps the reference's own `decode_output` call (bit-identical softmax; set
//!    gpu_threads_per_device >= 4 with it for a 256-channel chess network in f16).  Since round 5 the gather + softmax is
//!    the last step of the network's own launch (one launch per batch, no copy operations: 532k) and
//!  * a batch's host work before the launch (`encode_input` of every board, `move_to_index` of every available move) is
//!    shared with `KZ_HIP_PREP_THREADS` scoped helper threads when asked for (default 0: all of it on the executor
//!    thread, which is what the committed seam figures of the default were measured with on the C++ mirror; the mirror's
//!    helper is a persistent thread, the scoped threads here spawn per batch — a different cost model that has not been
//!    compiled or timed, so it is opt-in): `prepare` below.
//!
//! NOT compiled in this repository's CI (no cargo in the build image); written against the cited signatures.

use std::collections::VecDeque;
use std::borrow::Borrow;
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::fmt::{Debug, Formatter};
use std::marker::PhantomData;
use std::sync::Arc;

use board_game::board::Board;
use internal_iterator::InternalIterator;

use crate::mapping::bit_buffer::BitBuffer;
use crate::mapping::BoardMapper;
use crate::network::common::decode_output;
use crate::network::{Network, ZeroEvaluation};
use crate::zero::values::ZeroValuesPov;
use board_game::pov::ScalarPov;
use board_game::wdl::WDL;
use std::borrow::Cow;

#[repr(C)]
#[derive(Debug, Default, Copy, Clone)]
pub struct KzModelInfo {
    pub input_channels: i32,
    pub board_h: i32,
    pub board_w: i32,
    pub input_scalar_channels: i32,
    pub input_bool_channels: i32,
    pub policy_len: i32,
    pub tower_depth: i32,
    pub tower_channels: i32,
    pub policy_kind: i32,
    pub bits_bytes: i32,
    pub param_count: i64,
    pub flops_per_eval: f64,
}

/// What `kz_engine_create` would choose for a model (kz_model_plan): host logic, no GPU touched
#[repr(C)]
#[derive(Debug, Copy, Clone)]
pub struct KzPathPlan {
    pub tower_path: [c_char; 48],
    pub launches_per_batch: i32,
    pub reserved: [i32; 3],
}

pub const KZ_DTYPE_F32: c_int = 0;
pub const KZ_DTYPE_F16: c_int = 1;
/// f32 tensors and the same <= 1e-4 parity as KZ_DTYPE_F32, the tower's products as three f16 MFMAs on (hi, lo) pairs
pub const KZ_DTYPE_F32_SPLIT16: c_int = 2;
pub const KZ_ENGINE_SLOTS: usize = 4;

// The C ABI of include/kz_hip.h, declaration for declaration (tests/test_rust_shim_text.py compares the two files:
// every function of the header is bound here with the same name, arity and pointer-ness).  `kz_model` / `kz_engine`
// are opaque: `*mut c_void` / `*const c_void`.
#[link(name = "kzhip")]
extern "C" {
    fn kz_last_error() -> *const c_char;
    fn kz_device_count(count: *mut c_int) -> c_int;
    fn kz_device_pci_bus_id(device: c_int, buf: *mut c_char, len: usize) -> c_int;
    fn kz_model_load(path: *const c_char, out: *mut *mut c_void) -> c_int;
    fn kz_model_load_memory(blob: *const c_void, len: usize, out: *mut *mut c_void) -> c_int;
    fn kz_model_load_onnx(path: *const c_char, input_scalar_channels: c_int, out: *mut *mut c_void) -> c_int;
    fn kz_model_load_onnx_memory(blob: *const c_void, len: usize, input_scalar_channels: c_int, out: *mut *mut c_void) -> c_int;
    fn kz_model_free(model: *mut c_void);
    fn kz_model_get_info(model: *const c_void, out: *mut KzModelInfo) -> c_int;
    fn kz_engine_create(model: *const c_void, device: c_int, max_batch: c_int, dtype: c_int, out: *mut *mut c_void) -> c_int;
    fn kz_engine_destroy(engine: *mut c_void);
    fn kz_model_supports_dtype(model: *const c_void, dtype: c_int) -> c_int;
    fn kz_model_plan(model: *const c_void, max_batch: c_int, dtype: c_int, out: *mut KzPathPlan) -> c_int;
    fn kz_engine_max_batch(engine: *const c_void) -> c_int;
    fn kz_engine_eval_dense(engine: *mut c_void, input_nchw: *const f32, batch: c_int, scalars_out: *mut f32, policy_out: *mut f32) -> c_int;
    fn kz_engine_eval_packed(engine: *mut c_void, bits: *const u8, bits_stride: usize, scalars_in: *const f32, batch: c_int, scalars_out: *mut f32, policy_out: *mut f32) -> c_int;
    fn kz_engine_eval_packed_decoded(engine: *mut c_void, bits: *const u8, bits_stride: usize, scalars_in: *const f32, batch: c_int, move_offsets: *const i64, move_indices: *const i32, values_out: *mut f32, probs_out: *mut f32) -> c_int;
    // asynchronous pair, slot in [0, KZ_ENGINE_SLOTS): inputs are copied to pinned staging before submit returns
    fn kz_engine_submit_packed(engine: *mut c_void, slot: c_int, bits: *const u8, bits_stride: usize, scalars_in: *const f32, batch: c_int) -> c_int;
    fn kz_engine_wait(engine: *mut c_void, slot: c_int, scalars_out: *mut f32, policy_out: *mut f32) -> c_int;
    fn kz_engine_wait_view(engine: *mut c_void, slot: c_int, scalars_out: *mut *const f32, policy_out: *mut *const f32) -> c_int;
    fn kz_engine_submit_packed_decoded(engine: *mut c_void, slot: c_int, bits: *const u8, bits_stride: usize, scalars_in: *const f32, batch: c_int, move_offsets: *const i64, move_indices: *const i32) -> c_int;
    fn kz_engine_wait_decoded(engine: *mut c_void, slot: c_int, values_out: *mut *const f32, probs_out: *mut *const f32) -> c_int;
    // device-resident entry points and helpers (benchmarks and parity tests; the server does not need them)
    fn kz_engine_enqueue_packed_device(engine: *mut c_void, d_bits: *const c_void, bits_stride: usize, d_scalars_in: *const c_void, batch: c_int, d_scalars_out: *mut c_void, d_policy_out: *mut c_void) -> c_int;
    fn kz_engine_enqueue_dense_device(engine: *mut c_void, d_input_nchw: *const c_void, batch: c_int, d_scalars_out: *mut c_void, d_policy_out: *mut c_void) -> c_int;
    fn kz_engine_synchronize(engine: *mut c_void) -> c_int;
    fn kz_device_malloc(device: c_int, bytes: usize, out: *mut *mut c_void) -> c_int;
    fn kz_device_free(device: c_int, ptr: *mut c_void) -> c_int;
    fn kz_memcpy_h2d(device: c_int, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    fn kz_memcpy_d2h(device: c_int, dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    fn kz_device_synchronize(device: c_int) -> c_int;
    fn kz_engine_set_profiling(engine: *mut c_void, enable: c_int) -> c_int;
    fn kz_engine_kernel_time(engine: *mut c_void, prefix: *const c_char, total_ms: *mut f64, launches: *mut i64) -> c_int;
    fn kz_engine_tower_path(engine: *const c_void) -> *const c_char;
    fn kz_engine_launch_geometry(engine: *const c_void, batch: c_int, workgroups: *mut c_int, boards_per_workgroup: *mut c_int) -> c_int;
    fn kz_engine_read_activation(engine: *mut c_void, name: *const c_char, batch: c_int, out_nchw: *mut f32) -> c_int;
}

/// The reference panics on every executor error (`unwrap()` cudnn.rs:70,78); keep that behaviour.
fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(kz_last_error()) }.to_string_lossy().into_owned();
        panic!("kzhip: {}", msg);
    }
}

pub fn hip_device_count() -> usize {
    let mut n = 0;
    check(unsafe { kz_device_count(&mut n) });
    n as usize
}

/// What replaces `kn_cuda_sys::wrapper::handle::CudaDevice` at the server seam (`ZeroSpecialization::Device`,
/// INTEGRATION.md §1): a HIP device ordinal.  `CudaDevice::all()` (server.rs:49) -> `HipDevice::all()`,
/// `CudaDevice::new(d).unwrap()` (server.rs:51) -> `HipDevice::new(d)`.
#[derive(Debug, Copy, Clone, Eq, PartialEq, Hash)]
pub struct HipDevice(pub usize);

impl HipDevice {
    pub fn all() -> Vec<HipDevice> {
        (0..hip_device_count()).map(HipDevice).collect()
    }
    pub fn new(index: i32) -> HipDevice {
        assert!(index >= 0 && (index as usize) < hip_device_count(), "No HIP device {}", index);
        HipDevice(index as usize)
    }
    pub fn pci_bus_id(self) -> String {
        let mut buf = [0 as c_char; 64];
        check(unsafe { kz_device_pci_bus_id(self.0 as c_int, buf.as_mut_ptr(), buf.len()) });
        unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
    }
}

/// Arithmetic of the engine, a start-up choice (environment variable `KZ_HIP_DTYPE`, read once by
/// `HipSpecialization`).  The reference's executor is f32 only (`DTensor::F32`, cudnn.rs:73) and the stated parity is
/// 1e-4, so the DEFAULT is the fastest path that keeps it: `KZ_DTYPE_F32_SPLIT16` where the network's shape allows,
/// else `KZ_DTYPE_F32`.  `f16` (3 x faster again, tolerance 5e-3 of the output scale, range +-65504 with overflow
/// reported as an error) is opt-in.
#[derive(Debug, Copy, Clone, Eq, PartialEq)]
pub enum HipDtype {
    Parity,
    F32,
    F16,
}

impl HipDtype {
    pub fn from_env() -> HipDtype {
        match std::env::var("KZ_HIP_DTYPE").as_deref() {
            Err(_) | Ok("parity") | Ok("f32split16") => HipDtype::Parity,
            Ok("f32") => HipDtype::F32,
            Ok("f16") => HipDtype::F16,
            Ok(other) => panic!("KZ_HIP_DTYPE must be parity, f32 or f16, got '{}'", other),
        }
    }
    fn resolve(self, model: &HipModel) -> c_int {
        match self {
            HipDtype::F32 => KZ_DTYPE_F32,
            HipDtype::F16 => KZ_DTYPE_F16,
            HipDtype::Parity => {
                if unsafe { kz_model_supports_dtype(model.ptr, KZ_DTYPE_F32_SPLIT16) } == 1 {
                    KZ_DTYPE_F32_SPLIT16
                } else {
                    KZ_DTYPE_F32
                }
            }
        }
    }
}

/// Host-side parsed model: the `G` of `ZeroSpecialization` (what `Arc<Graph>` is for `AlphaZeroSpecialization`).
pub struct HipModel {
    ptr: *mut c_void,
    pub info: KzModelInfo,
}

// kz_model is immutable and thread-safe (include/kz_hip.h)
unsafe impl Send for HipModel {}
unsafe impl Sync for HipModel {}

impl HipModel {
    /// `path`: the ONNX file the trainer writes (python/lib/save_onnx.py); `input_scalar_count`: the mapper's, the
    /// graph itself does not say which of its input planes are broadcast scalars.
    pub fn load(path: &str, input_scalar_count: usize) -> Self {
        let c_path = CString::new(path).unwrap();
        let mut ptr = std::ptr::null_mut();
        check(unsafe { kz_model_load_onnx(c_path.as_ptr(), input_scalar_count as c_int, &mut ptr) });
        let mut info = KzModelInfo::default();
        check(unsafe { kz_model_get_info(ptr, &mut info) });
        HipModel { ptr, info }
    }
}

impl Drop for HipModel {
    fn drop(&mut self) {
        unsafe { kz_model_free(self.ptr) }
    }
}

pub struct HipNetwork<B: Board, M: BoardMapper<B>> {
    mapper: M,
    max_batch_size: usize,
    engine: *mut c_void,
    _model: Arc<HipModel>,

    bits: Vec<u8>,
    scalars_in: Vec<f32>,
    scalars_out: Vec<f32>,
    policy_out: Vec<f32>,
    /// boards of the batches in flight, oldest first (decode_output needs their available moves); with the
    /// device-side decode the boards are dropped at submit and the CSR offsets of their move lists are kept instead
    pending: VecDeque<(usize, Vec<B>, Vec<i64>)>,
    next_slot: usize,
    /// `KZ_HIP_DECODE` (default "device"): decode_output's gather + softmax on the GPU
    device_decode: bool,
    move_offsets: Vec<i64>,
    move_indices: Vec<i32>,
    /// `KZ_HIP_PREP_THREADS` (default 0) + 1 ranges of a batch, each prepared by one thread (`prepare`)
    ranges: Vec<PrepRange>,
    ph: PhantomData<B>,
}

/// What one thread prepares for its contiguous range of a batch's boards.
#[derive(Default)]
struct PrepRange {
    scalars: Vec<f32>,
    /// available moves per board
    counts: Vec<i64>,
    /// `move_to_index` of every available move, board after board
    indices: Vec<i32>,
}

/// `encode_input` of the boards of one range into `bits` (that range's part of the staging) and `out.scalars`, and — with
/// the decode on the device — `move_to_index` of every available move (what decode_output computes per board,
/// common.rs:77-81, hoisted in front of the evaluation).
fn prep_range<B: Board, M: BoardMapper<B>>(mapper: M, boards: &[&B], bits: &mut [u8], with_moves: bool, out: &mut PrepRange) {
    let bool_count = mapper.input_bool_len();
    let bits_bytes = (bool_count + 7) / 8;
    let policy_len = mapper.policy_len();
    out.scalars.clear();
    out.counts.clear();
    out.indices.clear();
    let mut buffer = BitBuffer::new(bool_count);
    for (bi, &board) in boards.iter().enumerate() {
        buffer.clear();
        // packed encode: exactly what BinaryOutput stores per position (binary_output.rs:210-256)
        mapper.encode_input(&mut buffer, &mut out.scalars, board);
        assert_eq!(bool_count, buffer.len());
        bits[bi * bits_bytes..(bi + 1) * bits_bytes].copy_from_slice(buffer.storage());
        if !with_moves {
            continue;
        }
        let before = out.indices.len();
        // `available_moves()` is an InternalIterator behind a Result (Err = the game is over: no moves, an empty
        // policy — decode_output's `map_or(vec![], ..)`, common.rs:77)
        if let Ok(moves) = board.available_moves() {
            let indices = &mut out.indices;
            moves.for_each(|mv| {
                let index = mapper.move_to_index(board, mv);
                assert!(index < policy_len);
                indices.push(index as i32);
            });
        }
        out.counts.push((out.indices.len() - before) as i64);
    }
}

// one engine per executor thread; it is moved into that thread once (executor.rs:320-342)
unsafe impl<B: Board, M: BoardMapper<B>> Send for HipNetwork<B, M> {}

impl<B: Board, M: BoardMapper<B>> HipNetwork<B, M> {
    /// Mirrors `CudaNetwork::new(mapper, &graph, max_batch_size, device)` (cudnn.rs:29-43).
    pub fn new(mapper: M, model: Arc<HipModel>, max_batch_size: usize, device: HipDevice, dtype: HipDtype) -> Self {
        // check_graph_shapes (network/common.rs:165-198)
        let info = model.info;
        let [c, h, w] = mapper.input_full_shape();
        assert_eq!(
            [info.input_channels as usize, info.board_h as usize, info.board_w as usize],
            [c, h, w],
            "Input shape mismatch between model and mapper"
        );
        assert_eq!(info.input_scalar_channels as usize, mapper.input_scalar_count());
        assert_eq!(info.policy_len as usize, mapper.policy_len(), "Wrong policy shape");

        let mut engine = std::ptr::null_mut();
        let dtype = dtype.resolve(&model);
        check(unsafe { kz_engine_create(model.ptr, device.0 as c_int, max_batch_size as c_int, dtype, &mut engine) });

        HipNetwork {
            mapper,
            max_batch_size,
            engine,
            bits: vec![0; max_batch_size * info.bits_bytes as usize],
            scalars_in: Vec::with_capacity(max_batch_size * mapper.input_scalar_count()),
            scalars_out: vec![0.0; max_batch_size * 5],
            policy_out: vec![0.0; max_batch_size * mapper.policy_len()],
            _model: model,
            pending: VecDeque::new(),
            next_slot: 0,
            device_decode: match std::env::var("KZ_HIP_DECODE").as_deref() {
                Err(_) | Ok("device") => true,
                Ok("host") => false,
                Ok(other) => panic!("KZ_HIP_DECODE must be device or host, got '{}'", other),
            },
            move_offsets: vec![],
            move_indices: vec![],
            ranges: {
                // (a malformed value is not worth a panic in a constructor: no helpers, and say so once)
                let helpers: usize = match std::env::var("KZ_HIP_PREP_THREADS") {
                    Err(_) => 0,
                    Ok(v) => v.trim().parse().unwrap_or_else(|_| {
                        eprintln!("kz_hip: KZ_HIP_PREP_THREADS='{}' is not a number: using 0 helper threads", v);
                        0
                    }),
                }
                .min(8);
                (0..helpers + 1).map(|_| PrepRange::default()).collect()
            },
            ph: PhantomData,
        }
    }

    /// A batch's host work before the launch — `encode_input` of every board (cudnn.rs:61-64) and, with the decode on the
    /// device, the CSR move lists `move_offsets[b]..move_offsets[b + 1]` of `move_indices` — cut into contiguous ranges:
    /// this thread prepares the first, a scoped helper thread each of the others (a thread spawn is ~15 us against the
    /// ~250 us half a chess batch of 256 takes); the ranges meet in the staging vectors in board order.  Returns
    /// bits_bytes.  Same results as one thread doing it all.
    fn prepare(&mut self, boards: &[impl Borrow<B>], with_moves: bool) -> usize {
        let mapper = self.mapper;
        let bits_bytes = (mapper.input_bool_len() + 7) / 8;
        // (`impl Borrow<B>` says nothing about threads; `&B` is Send + Sync because `Board` is)
        let refs: Vec<&B> = boards.iter().map(|b| b.borrow()).collect();
        let n = refs.len();
        let parts = self.ranges.len().min((n / 16).max(1)); // a handful of boards: not worth a spawn
        let cut = |k: usize| n * k / parts;
        let (first, rest) = self.ranges.split_at_mut(1);
        let (bits_first, mut bits_rest) = self.bits[..n * bits_bytes].split_at_mut(cut(1) * bits_bytes);
        std::thread::scope(|scope| {
            for (k, range) in rest[..parts - 1].iter_mut().enumerate() {
                let (lo, hi) = (cut(k + 1), cut(k + 2));
                let (mine, tail) = std::mem::take(&mut bits_rest).split_at_mut((hi - lo) * bits_bytes);
                bits_rest = tail;
                let part = &refs[lo..hi];
                scope.spawn(move || prep_range(mapper, part, mine, with_moves, range));
            }
            prep_range(mapper, &refs[..cut(1)], bits_first, with_moves, &mut first[0]);
            // (the scope joins the helpers; a panic in one of them — `move_to_index` asserts — propagates here)
        });
        self.scalars_in.clear();
        self.move_offsets.clear();
        self.move_offsets.push(0);
        self.move_indices.clear();
        for range in &self.ranges[..parts] {
            self.scalars_in.extend_from_slice(&range.scalars);
            if with_moves {
                self.move_indices.extend_from_slice(&range.indices);
                for &c in &range.counts {
                    self.move_offsets.push(self.move_offsets.last().unwrap() + c);
                }
            }
        }
        assert_eq!(self.scalars_in.len(), n * mapper.input_scalar_count());
        bits_bytes
    }

    /// values [n, 5] (tanh / softmax already applied on the device) + probabilities parallel to the move lists
    fn assemble(offsets: &[i64], values: &[f32], probs: &[f32]) -> Vec<ZeroEvaluation<'static>> {
        (0..offsets.len() - 1)
            .map(|bi| {
                let v = &values[bi * 5..bi * 5 + 5];
                let values = ZeroValuesPov {
                    value: ScalarPov::new(v[0]),
                    wdl: WDL { win: v[1], draw: v[2], loss: v[3] },
                    moves_left: v[4],
                };
                let policy = probs[offsets[bi] as usize..offsets[bi + 1] as usize].to_vec();
                ZeroEvaluation { values, policy: Cow::Owned(policy) }
            })
            .collect()
    }

    /// Asynchronous pair for `pipelined_executor_loop` (executor_pipelined.rs): encode, hand the batch to the engine
    /// and return while the GPU works.  At most `KZ_ENGINE_SLOTS` batches in flight.
    pub fn submit_batch(&mut self, boards: Vec<B>) {
        assert!(!boards.is_empty() && boards.len() <= self.max_batch_size);
        assert!(self.pending.len() < KZ_ENGINE_SLOTS, "every engine slot is in flight");
        let bits_bytes = self.prepare(&boards, self.device_decode);
        let slot = self.next_slot;
        if self.device_decode {
            check(unsafe {
                kz_engine_submit_packed_decoded(
                    self.engine,
                    slot as c_int,
                    self.bits.as_ptr(),
                    bits_bytes,
                    self.scalars_in.as_ptr(),
                    boards.len() as c_int,
                    self.move_offsets.as_ptr(),
                    self.move_indices.as_ptr(),
                )
            });
            // (the engine has copied the lists to its pinned staging: only the offsets are needed to cut up the reply)
            self.pending.push_back((slot, vec![], self.move_offsets.clone()));
        } else {
            check(unsafe {
                kz_engine_submit_packed(
                    self.engine,
                    slot as c_int,
                    self.bits.as_ptr(),
                    bits_bytes,
                    self.scalars_in.as_ptr(),
                    boards.len() as c_int,
                )
            });
            self.pending.push_back((slot, boards, vec![]));
        }
        self.next_slot = (slot + 1) % KZ_ENGINE_SLOTS;
    }

    /// Results of the OLDEST submitted batch.
    pub fn wait_batch(&mut self) -> Vec<ZeroEvaluation<'static>> {
        let (slot, boards, offsets) = self.pending.pop_front().expect("wait_batch with nothing in flight");
        if self.device_decode {
            let (mut values, mut probs) = (std::ptr::null::<f32>(), std::ptr::null::<f32>());
            check(unsafe { kz_engine_wait_decoded(self.engine, slot as c_int, &mut values, &mut probs) });
            let (n, total) = (offsets.len() - 1, *offsets.last().unwrap() as usize);
            // views into the engine's pinned staging, valid until the next submit on this slot (include/kz_hip.h)
            let values = unsafe { std::slice::from_raw_parts(values, n * 5) };
            let probs = if total == 0 { &[][..] } else { unsafe { std::slice::from_raw_parts(probs, total) } };
            return Self::assemble(&offsets, values, probs);
        }
        // decode straight from the engine's pinned staging (no copy of the 1.9 MB policy tensor into a Vec first)
        let (mut scalars, mut policy) = (std::ptr::null::<f32>(), std::ptr::null::<f32>());
        check(unsafe { kz_engine_wait_view(self.engine, slot as c_int, &mut scalars, &mut policy) });
        let (batch_size, policy_len) = (boards.len(), self.mapper.policy_len());
        let outputs = unsafe {
            [std::slice::from_raw_parts(scalars, batch_size * 5), std::slice::from_raw_parts(policy, batch_size * policy_len)]
        };
        decode_output(self.mapper, &boards, &outputs)
    }
}

impl<B: Board, M: BoardMapper<B>> Drop for HipNetwork<B, M> {
    fn drop(&mut self) {
        unsafe { kz_engine_destroy(self.engine) }
    }
}

impl<B: Board, M: BoardMapper<B>> Network<B> for HipNetwork<B, M> {
    fn max_batch_size(&self) -> usize {
        self.max_batch_size
    }

    fn evaluate_batch(&mut self, boards: &[impl Borrow<B>]) -> Vec<ZeroEvaluation<'static>> {
        let batch_size = boards.len();
        assert!(batch_size <= self.max_batch_size);
        if batch_size == 0 {
            return vec![];
        }

        assert!(self.pending.is_empty(), "evaluate_batch while submitted batches are in flight");
        let bits_bytes = self.prepare(boards, self.device_decode);

        if self.device_decode {
            check(unsafe {
                kz_engine_submit_packed_decoded(
                    self.engine,
                    0,
                    self.bits.as_ptr(),
                    bits_bytes,
                    self.scalars_in.as_ptr(),
                    batch_size as c_int,
                    self.move_offsets.as_ptr(),
                    self.move_indices.as_ptr(),
                )
            });
            let (mut values, mut probs) = (std::ptr::null::<f32>(), std::ptr::null::<f32>());
            check(unsafe { kz_engine_wait_decoded(self.engine, 0, &mut values, &mut probs) });
            let total = *self.move_offsets.last().unwrap() as usize;
            let values = unsafe { std::slice::from_raw_parts(values, batch_size * 5) };
            let probs = if total == 0 { &[][..] } else { unsafe { std::slice::from_raw_parts(probs, total) } };
            return Self::assemble(&self.move_offsets, values, probs);
        }

        check(unsafe {
            kz_engine_eval_packed(
                self.engine,
                self.bits.as_ptr(),
                bits_bytes,
                self.scalars_in.as_ptr(),
                batch_size as c_int,
                self.scalars_out.as_mut_ptr(),
                self.policy_out.as_mut_ptr(),
            )
        });

        let policy_len = self.mapper.policy_len();
        let outputs = [&self.scalars_out[..batch_size * 5], &self.policy_out[..batch_size * policy_len]];
        decode_output(self.mapper, boards, &outputs)
    }
}

impl<B: Board, M: BoardMapper<B>> Debug for HipNetwork<B, M> {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("HipNetwork")
            .field("mapper", &self.mapper)
            .field("max_batch_size", &self.max_batch_size)
            .finish()
    }
}

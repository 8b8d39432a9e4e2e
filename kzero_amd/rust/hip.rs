//! kz-core/src/network/hip.rs — `HipNetwork`: the MI355X executor behind kZero's `Network` trait.
//!
//! Drop-in sibling of `CudaNetwork` (kz-core/src/network/cudnn.rs:18-88): same constructor shape, same
//! `Network<B>` contract (kz-core/src/network/mod.rs:52-63), same `decode_output` (network/common.rs:16-100).
//! The arithmetic lives in libkzhip.so (C ABI: include/kz_hip.h); this file only binds it.
//!
//! Differences from `CudaNetwork`, all invisible to callers:
//!  * the mapper's *packed* output (`InputMapper::encode_input`: BitBuffer + scalars, mapping/mod.rs:37) is handed
//!    over as is; the dense f32 expansion (`encode_input_full`, mapping/mod.rs:40-63) happens on the GPU;
//!  * no NaN padding to `max_batch_size` (cudnn.rs:65) and no input clone (cudnn.rs:70): only `batch` rows exist.
//!
//! NOT compiled in this repository's CI (no cargo in the build image); written against the cited signatures.

use std::collections::VecDeque;
use std::borrow::Borrow;
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::fmt::{Debug, Formatter};
use std::marker::PhantomData;
use std::sync::Arc;

use board_game::board::Board;

use crate::mapping::bit_buffer::BitBuffer;
use crate::mapping::BoardMapper;
use crate::network::common::decode_output;
use crate::network::{Network, ZeroEvaluation};

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

pub const KZ_DTYPE_F32: c_int = 0;
pub const KZ_DTYPE_F16: c_int = 1;
/// f32 tensors and the same <= 1e-4 parity as KZ_DTYPE_F32, the tower's products as three f16 MFMAs on (hi, lo) pairs
pub const KZ_DTYPE_F32_SPLIT16: c_int = 2;

#[link(name = "kzhip")]
extern "C" {
    fn kz_last_error() -> *const c_char;
    fn kz_device_count(count: *mut c_int) -> c_int;
    fn kz_model_load_onnx(path: *const c_char, input_scalar_channels: c_int, out: *mut *mut c_void) -> c_int;
    fn kz_model_free(model: *mut c_void);
    fn kz_model_get_info(model: *const c_void, out: *mut KzModelInfo) -> c_int;
    fn kz_engine_create(model: *const c_void, device: c_int, max_batch: c_int, dtype: c_int, out: *mut *mut c_void) -> c_int;
    fn kz_engine_destroy(engine: *mut c_void);
    fn kz_engine_eval_packed(
        engine: *mut c_void,
        bits: *const u8,
        bits_stride: usize,
        scalars_in: *const f32,
        batch: c_int,
        scalars_out: *mut f32,
        policy_out: *mut f32,
    ) -> c_int;
    // asynchronous pair, slot in [0, KZ_ENGINE_SLOTS): inputs are copied to pinned staging before submit returns
    fn kz_engine_submit_packed(
        engine: *mut c_void,
        slot: c_int,
        bits: *const u8,
        bits_stride: usize,
        scalars_in: *const f32,
        batch: c_int,
    ) -> c_int;
    fn kz_engine_wait(engine: *mut c_void, slot: c_int, scalars_out: *mut f32, policy_out: *mut f32) -> c_int;
}

pub const KZ_ENGINE_SLOTS: usize = 2;

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
    /// boards of the batches in flight, oldest first (decode_output needs their available moves)
    pending: VecDeque<(usize, Vec<B>)>,
    next_slot: usize,
    ph: PhantomData<B>,
}

// one engine per executor thread; it is moved into that thread once (executor.rs:320-342)
unsafe impl<B: Board, M: BoardMapper<B>> Send for HipNetwork<B, M> {}

impl<B: Board, M: BoardMapper<B>> HipNetwork<B, M> {
    /// Mirrors `CudaNetwork::new(mapper, &graph, max_batch_size, device)` (cudnn.rs:29-43).
    pub fn new(mapper: M, model: Arc<HipModel>, max_batch_size: usize, device: usize, dtype: c_int) -> Self {
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
        check(unsafe { kz_engine_create(model.ptr, device as c_int, max_batch_size as c_int, dtype, &mut engine) });

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
            ph: PhantomData,
        }
    }

    /// packed encode: exactly what BinaryOutput stores per position (binary_output.rs:210-256)
    fn encode_into_staging(&mut self, boards: &[impl Borrow<B>]) -> usize {
        let bool_count = self.mapper.input_bool_len();
        let bits_bytes = (bool_count + 7) / 8;
        self.scalars_in.clear();
        let mut buffer = BitBuffer::new(bool_count);
        for (bi, board) in boards.iter().enumerate() {
            buffer.clear();
            self.mapper.encode_input(&mut buffer, &mut self.scalars_in, board.borrow());
            assert_eq!(bool_count, buffer.len());
            self.bits[bi * bits_bytes..(bi + 1) * bits_bytes].copy_from_slice(buffer.storage());
        }
        assert_eq!(self.scalars_in.len(), boards.len() * self.mapper.input_scalar_count());
        bits_bytes
    }

    /// Asynchronous pair for `pipelined_executor_loop` (executor_pipelined.rs): encode, hand the batch to the engine
    /// and return while the GPU works.  At most `KZ_ENGINE_SLOTS` batches in flight.
    pub fn submit_batch(&mut self, boards: Vec<B>) {
        assert!(!boards.is_empty() && boards.len() <= self.max_batch_size);
        assert!(self.pending.len() < KZ_ENGINE_SLOTS, "every engine slot is in flight");
        let bits_bytes = self.encode_into_staging(&boards);
        let slot = self.next_slot;
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
        self.pending.push_back((slot, boards));
        self.next_slot = (slot + 1) % KZ_ENGINE_SLOTS;
    }

    /// Results of the OLDEST submitted batch.
    pub fn wait_batch(&mut self) -> Vec<ZeroEvaluation<'static>> {
        let (slot, boards) = self.pending.pop_front().expect("wait_batch with nothing in flight");
        check(unsafe {
            kz_engine_wait(self.engine, slot as c_int, self.scalars_out.as_mut_ptr(), self.policy_out.as_mut_ptr())
        });
        let (batch_size, policy_len) = (boards.len(), self.mapper.policy_len());
        let outputs = [&self.scalars_out[..batch_size * 5], &self.policy_out[..batch_size * policy_len]];
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
        let bits_bytes = self.encode_into_staging(boards);

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

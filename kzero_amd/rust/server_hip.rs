//! kz-selfplay/src/server/server_hip.rs — `HipSpecialization`: the MI355X executor behind the server's
//! `ZeroSpecialization` seam (kz-selfplay/src/server/server.rs:287-302).
//!
//! The seam itself carries a kn-cuda-sys type today (`device: CudaDevice`, server.rs:293; `devices: Vec<CudaDevice>`,
//! server.rs:105,207,249,306; enumeration server.rs:47-53), so it is edited, once, to stop naming a backend
//! (INTEGRATION.md §1 lists every line):
//!
//! ```ignore
//! pub trait ZeroSpecialization<B: Board, M: BoardMapper<B> + 'static> {
//!     type G: Send + Sync;
//!     type Device: Copy + Debug + Send + Sync + 'static;                 // NEW
//!     /// `indices` = the `--device` arguments; empty = every device      // NEW (was server.rs:48-52)
//!     fn devices(indices: &[i32]) -> Vec<Self::Device>;
//!     fn spawn_device_threads<'s>(&self, s: &Scope<'s>, device: Self::Device, /* the rest unchanged */ ...)
//!         -> (Vec<Sender<Settings>>, Vec<GraphSender<Self::G>>);
//!     fn load_graph(&self, path: &str, mapper: M, startup: &StartupSettings) -> Self::G;
//! }
//! ```
//!
//! `AlphaZeroSpecialization` / `MuZeroSpecialization` get `type Device = CudaDevice;` and the moved enumeration;
//! `selfplay_server_main` and the three dispatch functions pass `args.device: Vec<i32>` down instead of
//! `Vec<CudaDevice>`, and `selfplay_start` begins with `let devices = Z::devices(&devices);`.
//!
//! No line of the reference is duplicated here.  The body of `AlphaZeroSpecialization::spawn_device_threads`
//! (`server_alphazero.rs:39-123`: sizing, `job_pair`, `ThreadPoolBuilder`, the `generator_alphazero_main` tasks, the
//! executor threads around `batched_executor_loop`) moves as it stands into a helper that is generic over the one thing
//! the two backends differ in — how an executor thread turns the graph it was sent into a `Network`:
//!
//! ```ignore
//! pub(super) fn spawn_alphazero_device_threads<'s, B, M, G, N>(
//!     s: &Scope<'s>, device_id: usize, startup: &StartupSettings, mapper: M,
//!     start_pos: impl Fn(&mut StdRng) -> B + Send + Sync + Clone + 'static,
//!     update_sender: Sender<GeneratorUpdate<B>>,
//!     load_network: impl Fn(Arc<G>) -> N + Send + Clone + 'static,      // the only new parameter
//! ) -> (Vec<Sender<Settings>>, Vec<GraphSender<G>>)
//! where B: Board + Hash, M: BoardMapper<B> + 'static, G: Send + Sync + 'static, N: Network<B>
//! {
//!     /* server_alphazero.rs:39-123 unchanged, except :106 */
//!     //     let inner = CudaNetwork::new(mapper, &graph, gpu_batch_size, device);     becomes
//!     //     let inner = load_network(graph);
//! }
//! ```
//!
//! `AlphaZeroSpecialization::spawn_device_threads` becomes the call
//! `spawn_alphazero_device_threads(s, device_id, startup, mapper, start_pos, update_sender,
//!     move |graph: Arc<Graph>| CudaNetwork::new(mapper, &graph, gpu_batch_size, device))`,
//! and this file supplies the same call with the HIP constructor, plus the model loader and the device list.
//!
//! Generators, MCTS, job channel, `batched_executor_loop`, collector, commander and the wire protocol are untouched.
//! NOT compiled in this repository's CI (no cargo in the build image); written against the cited signatures and kept
//! consistent with include/kz_hip.h by tests/test_rust_shim_text.py.

use std::hash::Hash;
use std::sync::Arc;

use board_game::board::Board;
use crossbeam::thread::Scope;
use flume::Sender;
use rand::rngs::StdRng;

use kz_core::mapping::BoardMapper;
use kz_core::network::hip::{HipDevice, HipDtype, HipModel, HipNetwork};

use crate::server::protocol::{GeneratorUpdate, Settings, StartupSettings};
use crate::server::server::{GraphSender, ZeroSpecialization};
use crate::server::server_alphazero::spawn_alphazero_device_threads;

#[derive(Debug)]
pub struct HipSpecialization {
    /// `KZ_HIP_DTYPE`: parity (default: <= 1e-4, the reference's f32 results) | f32 | f16
    pub dtype: HipDtype,
}

impl HipSpecialization {
    pub fn from_env() -> Self {
        HipSpecialization { dtype: HipDtype::from_env() }
    }
}

impl<B: Board + Hash, M: BoardMapper<B> + 'static> ZeroSpecialization<B, M> for HipSpecialization {
    /// What the commander wraps in an `Arc` and clones to every executor (commander.rs:36-45).
    type G = HipModel;
    type Device = HipDevice;

    /// server.rs:47-53 for this backend.
    fn devices(indices: &[i32]) -> Vec<HipDevice> {
        let devices = if indices.is_empty() {
            HipDevice::all()
        } else {
            indices.iter().map(|&d| HipDevice::new(d)).collect()
        };
        assert!(!devices.is_empty(), "No HIP devices found");
        devices
    }

    fn spawn_device_threads<'s>(
        &self,
        s: &Scope<'s>,
        device: HipDevice,
        device_id: usize,
        startup: &StartupSettings,
        mapper: M,
        start_pos: impl Fn(&mut StdRng) -> B + Send + Sync + Clone + 'static,
        update_sender: Sender<GeneratorUpdate<B>>,
    ) -> (Vec<Sender<Settings>>, Vec<GraphSender<HipModel>>) {
        let (gpu_batch_size, dtype) = (startup.gpu_batch_size, self.dtype);
        // one engine per executor thread, created on that thread by `handle_new_graph` (executor.rs:320-342); engines of
        // one device share the uploaded weights inside libkzhip
        spawn_alphazero_device_threads(s, device_id, startup, mapper, start_pos, update_sender, move |model: Arc<HipModel>| {
            HipNetwork::new(mapper, model, gpu_batch_size, device, dtype)
        })
    }

    /// The ONNX path of `Command::NewNetwork` (protocol.rs:36) goes straight to the C ABI; the mapper supplies the one
    /// fact the graph does not carry (how many input planes are broadcast scalars).
    fn load_graph(&self, path: &str, mapper: M, _: &StartupSettings) -> HipModel {
        HipModel::load(path, mapper.input_scalar_count())
    }
}

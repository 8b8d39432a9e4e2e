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
//! Generator half: `server_alphazero.rs:39-87` (sizing, `job_pair`, `ThreadPoolBuilder`, the
//! `generator_alphazero_main` tasks) moves unchanged into
//! `pub(super) fn spawn_alphazero_generators(s, device_id, startup, mapper, start_pos, update_sender)
//!     -> (Vec<Sender<Settings>>, JobServer<B, ZeroEvaluation<'static>>)`,
//! which `AlphaZeroSpecialization::spawn_device_threads` keeps calling first.  This file supplies the other half for
//! the HIP backend: the executor threads and the model loader.
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
use rand::thread_rng;

use kz_core::mapping::BoardMapper;
use kz_core::network::hip::{HipDevice, HipDtype, HipModel, HipNetwork};
use kz_core::network::symmetry::RandomSymmetryNetwork;
use kz_core::network::Network;

use crate::server::executor::{batched_executor_loop, RunCondition};
use crate::server::protocol::{Evals, GeneratorUpdate, Settings, StartupSettings};
use crate::server::server::{GraphSender, ZeroSpecialization};
use crate::server::server_alphazero::spawn_alphazero_generators;

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
        let (settings_senders, eval_server) =
            spawn_alphazero_generators(s, device_id, startup, mapper, start_pos, update_sender.clone());

        let gpu_batch_size = startup.gpu_batch_size;
        let eval_job_count = gpu_batch_size / startup.search_batch_size; // server_alphazero.rs:48
        let eval_random_symmetries = startup.eval_random_symmetries;
        let dtype = self.dtype;

        let mut graph_senders: Vec<GraphSender<HipModel>> = vec![];
        // spawn gpu eval threads (server_alphazero.rs:89-121 with the network constructor exchanged)
        for local_id in 0..startup.gpu_threads_per_device {
            let (graph_sender, graph_receiver) = flume::bounded(1);
            graph_senders.push(graph_sender);

            let eval_server = eval_server.clone();
            let update_sender = update_sender.clone();

            s.builder()
                .name(format!("gpu-expand-{}-{}", device_id, local_id))
                .spawn(move |_| {
                    batched_executor_loop(
                        gpu_batch_size,
                        RunCondition::JobCount(eval_job_count),
                        graph_receiver,
                        eval_server,
                        // one engine per executor thread, created on that thread (executor.rs:320-342); engines of
                        // one device share the uploaded weights inside libkzhip
                        |graph| {
                            graph.map_left(|model: Arc<HipModel>| {
                                let inner = HipNetwork::new(mapper, model, gpu_batch_size, device, dtype);
                                RandomSymmetryNetwork::new(inner, thread_rng(), eval_random_symmetries)
                            })
                        },
                        |network, x| {
                            let y = network.evaluate_batch(&x);
                            // the collector's `real` evals/s (server_alphazero.rs:111-117)
                            let msg =
                                GeneratorUpdate::ExpandEvals(Evals::new(x.len() as u64, gpu_batch_size as u64, 0));
                            update_sender.send(msg).unwrap();
                            y
                        },
                    );
                })
                .unwrap();
        }

        (settings_senders, graph_senders)
    }

    /// The ONNX path of `Command::NewNetwork` (protocol.rs:36) goes straight to the C ABI; the mapper supplies the one
    /// fact the graph does not carry (how many input planes are broadcast scalars).
    fn load_graph(&self, path: &str, mapper: M, _: &StartupSettings) -> HipModel {
        HipModel::load(path, mapper.input_scalar_count())
    }
}

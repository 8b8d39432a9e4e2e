//! kz-selfplay/src/server/server_hip.rs — `HipSpecialization`: the MI355X executor behind the server's
//! `ZeroSpecialization` seam (kz-selfplay/src/server/server.rs:287-302).
//!
//! What a maintainer changes, and nothing else:
//!   1. In `server_alphazero.rs`, the body of `spawn_device_threads` up to and including the generator pool
//!      (lines 39-87: sizing, `job_pair`, `ThreadPoolBuilder`, `generator_alphazero_main` tasks) moves unchanged into a
//!      helper `spawn_alphazero_generators(device_id, startup, start_pos, update_sender) ->
//!      (Vec<Sender<Settings>>, JobServer<B, ZeroEvaluation<'static>>)`, which `AlphaZeroSpecialization` keeps calling.
//!   2. This file supplies the other half for the HIP backend: the executor threads and the model loader.
//!
//! Generators, MCTS, job channel, `batched_executor_loop`, collector, commander and the wire protocol are untouched.
//! NOT compiled in this repository's CI (no cargo in the build image); written against the cited signatures.

use std::sync::Arc;

use board_game::board::Board;
use crossbeam::thread::Scope;
use flume::Sender;
use rand::rngs::StdRng;
use rand::thread_rng;

use kz_core::mapping::BoardMapper;
use kz_core::network::hip::{HipModel, HipNetwork, KZ_DTYPE_F16};
use kz_core::network::symmetry::RandomSymmetryNetwork;
use kz_core::network::Network;

use crate::server::executor::{batched_executor_loop, RunCondition};
use crate::server::protocol::{Evals, GeneratorUpdate, Settings, StartupSettings};
use crate::server::server::{GraphSender, ZeroSpecialization};
use crate::server::server_alphazero::spawn_alphazero_generators; // step 1 above

#[derive(Debug)]
pub struct HipSpecialization;

impl<B: Board + std::hash::Hash, M: BoardMapper<B> + 'static> ZeroSpecialization<B, M> for HipSpecialization {
    /// What the commander wraps in an `Arc` and clones to every executor (commander.rs:36-45).
    type G = HipModel;

    fn spawn_device_threads<'s>(
        &self,
        s: &Scope<'s>,
        device: CudaDevice, // only its index is used; with the cuda crates gone this is a plain device index
        device_id: usize,
        startup: &StartupSettings,
        mapper: M,
        start_pos: impl Fn(&mut StdRng) -> B + Send + Sync + Clone + 'static,
        update_sender: Sender<GeneratorUpdate<B>>,
    ) -> (Vec<Sender<Settings>>, Vec<GraphSender<HipModel>>) {
        let (settings_senders, eval_server) =
            spawn_alphazero_generators(device_id, startup, start_pos, update_sender.clone());

        let batch = startup.gpu_batch_size;
        let jobs_per_batch = batch / startup.search_batch_size; // RunCondition::JobCount, as before
        let symmetries = startup.eval_random_symmetries;
        let device_index = device.inner() as usize;

        let graph_senders = (0..startup.gpu_threads_per_device)
            .map(|local_id| {
                let (graph_sender, graph_receiver) = flume::bounded(1);
                let (eval_server, update_sender) = (eval_server.clone(), update_sender.clone());
                s.builder()
                    .name(format!("gpu-expand-{}-{}", device_id, local_id))
                    .spawn(move |_| {
                        batched_executor_loop(
                            batch,
                            RunCondition::JobCount(jobs_per_batch),
                            graph_receiver,
                            eval_server,
                            // one engine per executor thread, created on that thread; engines of one device share
                            // the uploaded weights inside libkzhip
                            |message| {
                                message.map_left(|model: Arc<HipModel>| {
                                    let engine = HipNetwork::new(mapper, model, batch, device_index, KZ_DTYPE_F16);
                                    RandomSymmetryNetwork::new(engine, thread_rng(), symmetries)
                                })
                            },
                            |network, boards| {
                                let evals = network.evaluate_batch(&boards);
                                let real = boards.len() as u64; // the collector's `real` evals/s
                                update_sender
                                    .send(GeneratorUpdate::ExpandEvals(Evals::new(real, batch as u64, 0)))
                                    .unwrap();
                                evals
                            },
                        )
                    })
                    .unwrap();
                graph_sender
            })
            .collect();

        (settings_senders, graph_senders)
    }

    /// The ONNX path of `Command::NewNetwork` (protocol.rs:36) goes straight to the C ABI; the mapper supplies the one
    /// fact the graph does not carry (how many input planes are broadcast scalars).
    fn load_graph(&self, path: &str, mapper: M, _: &StartupSettings) -> HipModel {
        HipModel::load(path, mapper.input_scalar_count())
    }
}

//! kz-selfplay/src/server/server_hip.rs — `HipSpecialization`: per-device construction for the MI355X executor.
//!
//! Same shape as `AlphaZeroSpecialization` (kz-selfplay/src/server/server_alphazero.rs:29-128) behind the same seam
//! (`trait ZeroSpecialization`, server.rs:287-302).  Only two things change: `type G = HipModel` (was `Graph`) and the
//! `load_network` closure builds a `HipNetwork` (was `CudaNetwork`).  Generators, job channel, `batched_executor_loop`,
//! collector, commander and the wire protocol are untouched.
//!
//! `CudaDevice` in the trait signature is only used for its index; with the cuda crates removed it becomes a plain
//! `usize` newtype.  NOT compiled in this repository's CI (no cargo in the build image).

use std::hash::Hash;
use std::sync::Arc;

use board_game::board::Board;
use crossbeam::thread::Scope;
use flume::Sender;
use futures::executor::ThreadPoolBuilder;
use rand::rngs::StdRng;
use rand::thread_rng;

use kz_core::mapping::BoardMapper;
use kz_core::network::hip::{HipModel, HipNetwork, KZ_DTYPE_F16};
use kz_core::network::job_channel::job_pair;
use kz_core::network::symmetry::RandomSymmetryNetwork;
use kz_core::network::Network;
use kz_util::math::ceil_div;

use crate::server::executor::{batched_executor_loop, RunCondition};
use crate::server::generator_alphazero::generator_alphazero_main;
use crate::server::protocol::{Evals, GeneratorUpdate, Settings, StartupSettings};
use crate::server::server::{GraphSender, ZeroSpecialization};

#[derive(Debug)]
pub struct HipSpecialization;

impl<B: Board + Hash, M: BoardMapper<B> + 'static> ZeroSpecialization<B, M> for HipSpecialization {
    type G = HipModel;

    fn spawn_device_threads<'s>(
        &self,
        s: &Scope<'s>,
        device: CudaDevice, // index only
        device_id: usize,
        startup: &StartupSettings,
        mapper: M,
        start_pos: impl Fn(&mut StdRng) -> B + Send + Sync + Clone + 'static,
        update_sender: Sender<GeneratorUpdate<B>>,
    ) -> (Vec<Sender<Settings>>, Vec<GraphSender<HipModel>>) {
        let gpu_batch_size = startup.gpu_batch_size;
        let search_batch_size = startup.search_batch_size;
        let cpu_threads = startup.cpu_threads_per_device;
        let gpu_threads = startup.gpu_threads_per_device;

        // identical sizing to server_alphazero.rs:47-55
        let concurrent_games = ceil_div((gpu_threads + 1) * gpu_batch_size, search_batch_size);
        let eval_job_count = gpu_batch_size / search_batch_size;
        let job_buffer_size = ceil_div(gpu_threads * gpu_batch_size, search_batch_size);
        let (eval_client, eval_server) = job_pair(job_buffer_size);

        let mut settings_senders = vec![];
        let mut graph_senders = vec![];

        let pool = ThreadPoolBuilder::new()
            .pool_size(cpu_threads)
            .name_prefix(format!("generator-{}-", device_id))
            .create()
            .unwrap();
        for local_generator_id in 0..concurrent_games {
            let generator_id = concurrent_games * device_id + local_generator_id;
            let (start_pos, eval_client, update_sender) = (start_pos.clone(), eval_client.clone(), update_sender.clone());
            let (settings_sender, settings_receiver) = flume::bounded(1);
            settings_senders.push(settings_sender);
            pool.spawn_ok(async move {
                generator_alphazero_main(generator_id, start_pos, settings_receiver, search_batch_size, eval_client, update_sender).await;
            });
        }

        for local_id in 0..gpu_threads {
            let (graph_sender, graph_receiver) = flume::bounded(1);
            graph_senders.push(graph_sender);
            let (eval_server, update_sender) = (eval_server.clone(), update_sender.clone());
            let eval_random_symmetries = startup.eval_random_symmetries;
            let device_index = device.inner() as usize;

            s.builder()
                .name(format!("gpu-expand-{}-{}", device_id, local_id))
                .spawn(move |_| {
                    batched_executor_loop(
                        gpu_batch_size,
                        RunCondition::JobCount(eval_job_count),
                        graph_receiver,
                        eval_server,
                        |model| {
                            model.map_left(|model: Arc<HipModel>| {
                                // one engine per executor thread; engines of one device share the uploaded weights
                                let inner = HipNetwork::new(mapper, model, gpu_batch_size, device_index, KZ_DTYPE_F16);
                                RandomSymmetryNetwork::new(inner, thread_rng(), eval_random_symmetries)
                            })
                        },
                        |network, x| {
                            let y = network.evaluate_batch(&x);
                            let msg = GeneratorUpdate::ExpandEvals(Evals::new(x.len() as u64, gpu_batch_size as u64, 0));
                            update_sender.send(msg).unwrap();
                            y
                        },
                    );
                })
                .unwrap();
        }

        (settings_senders, graph_senders)
    }

    /// Replaces `optimize_graph(&load_graph_from_onnx_path(path, false)?, ..)` (server_alphazero.rs:126-128): the same
    /// ONNX path the commander receives in `Command::NewNetwork` (protocol.rs:36) goes straight to the C ABI.
    fn load_graph(&self, path: &str, mapper: M, _: &StartupSettings) -> HipModel {
        HipModel::load(path, mapper.input_scalar_count())
    }
}

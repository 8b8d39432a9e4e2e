//! Addition to kz-selfplay/src/server/executor.rs (same module: it uses `State`, `RunCondition`, `Message` and
//! `handle_new_graph` as they are) — SURVEY.md §8(f) N3: the executor loop with up to `depth` batches in flight on ONE
//! thread, for a network with an asynchronous pair `submit(&mut N, Vec<X>)` / `wait(&mut N) -> Vec<Y>` (results of the
//! OLDEST submitted batch): `HipNetwork::{submit_batch, wait_batch}` over `kz_engine_submit_packed` / `kz_engine_wait`.
//!
//! Channel semantics are those of `batched_executor_loop` (executor.rs:27-146): the same `RunCondition`, replies in job
//! order, a new graph first drains what is in flight on the old network, a disconnected job channel evaluates the
//! remainder and returns.  What changes is only when the thread blocks: while the GPU works on a batch the thread
//! keeps collecting jobs, encodes and submits the next batch; it blocks on the channels only when nothing is in flight
//! and on the GPU only when it cannot submit.  One such thread replaces `gpu_threads_per_device` blocking threads
//! (Readme.md:51).  C++ mirror with ASan/TSan tests: kzero_amd/csrc/host/executor.hpp, tests/cpp/test_host.cpp.
//! NOT compiled in this repository's CI (no cargo in the build image); written against the cited signatures.

/// What `State` needs on top of executor.rs:176-302: the inputs of a submitted batch leave `x` (the network owns them
/// until the results come back) while their senders stay queued, so
///   * `check_invariants` becomes `items_to_eval() + in_flight_items + items_waiting_for_send() == items_to_send()`,
///   * `RunCondition::JobCount(n)` counts the jobs that still have unsubmitted inputs: `senders.len() - covered_jobs`.
struct Pipeline {
    /// sizes of the batches in flight, oldest first
    in_flight: VecDeque<usize>,
    /// running index of the last input of every queued job / number of inputs submitted so far
    job_ends: VecDeque<u64>,
    pushed_total: u64,
    submitted_total: u64,
}

impl Pipeline {
    fn pending_jobs(&self) -> usize {
        self.job_ends.len()
    }
    fn note_pushed(&mut self, len: usize) {
        if len > 0 {
            self.pushed_total += len as u64;
            self.job_ends.push_back(self.pushed_total);
        }
    }
    fn note_submitted(&mut self, len: usize) {
        self.submitted_total += len as u64;
        while self.job_ends.front().map_or(false, |&end| end <= self.submitted_total) {
            self.job_ends.pop_front();
        }
        self.in_flight.push_back(len);
    }
}

pub fn pipelined_executor_loop<G, N, X, Y>(
    max_batch_size: usize,
    depth: usize,
    run_condition: RunCondition,
    graph_receiver: Receiver<Option<G>>,
    server: JobServer<X, Y>,
    mut load_network: impl FnMut(G) -> N,
    mut submit_batch: impl FnMut(&mut N, Vec<X>),
    mut wait_batch: impl FnMut(&mut N) -> Vec<Y>,
) {
    let thread_name = std::thread::current().name().unwrap_or("unnamed").to_owned();
    assert_ne!(max_batch_size, 0, "Got batch size 0 for {}", thread_name);
    assert_ne!(depth, 0, "Got pipeline depth 0 for {}", thread_name);

    let job_receiver = server.into_receiver();
    let mut state = State::new();
    let mut pipe = Pipeline { in_flight: VecDeque::new(), job_ends: VecDeque::new(), pushed_total: 0, submitted_total: 0 };
    let mut network: Option<N> = None;
    let (mut graph_disconnected, mut jobs_disconnected) = (false, false);

    // should_eval (executor.rs:240-253) with JobCount over the jobs not yet handed to the network
    let ready = |state: &State<X, Y>, pipe: &Pipeline, jobs_disconnected: bool| -> bool {
        let pending = state.x.len();
        if pending == 0 {
            return false;
        }
        if jobs_disconnected || pending >= max_batch_size {
            return true;
        }
        match run_condition {
            RunCondition::FullBatch => false,
            RunCondition::JobCount(count) => pipe.pending_jobs() >= count,
            RunCondition::Any => true,
        }
    };

    macro_rules! submit_one {
        () => {{
            let n = min(state.x.len(), max_batch_size);
            let batch_x: Vec<X> = state.x.drain(0..n).collect();
            begin_event_with_color("run", CL_GREEN);
            submit_batch(network.as_mut().unwrap(), batch_x);
            end_event();
            pipe.note_submitted(n);
        }};
    }
    macro_rules! finish_oldest {
        () => {{
            let batch_y = wait_batch(network.as_mut().unwrap());
            assert_eq!(Some(batch_y.len()), pipe.in_flight.pop_front());
            begin_event_with_color("reply", CL_YELLOW);
            respond_in_flight(&mut state, batch_y);
            end_event();
        }};
    }
    // non-blocking: everything already queued, up to a full batch beyond what is in flight (executor.rs:83-92)
    macro_rules! drain_jobs {
        () => {{
            while state.x.len() < max_batch_size {
                match job_receiver.try_recv() {
                    Ok(job) => {
                        pipe.note_pushed(job.x.len());
                        state.push_job_pipelined(job);
                    }
                    Err(TryRecvError::Empty) => break,
                    Err(TryRecvError::Disconnected) => {
                        jobs_disconnected = true;
                        break;
                    }
                }
            }
        }};
    }

    loop {
        assert!(network.is_some() || !graph_disconnected);

        if !pipe.in_flight.is_empty() {
            // The GPU is busy: never block on the channels.  A new graph takes effect between batches, as in the
            // synchronous loop (there it is seen once the running evaluate_batch returns).
            if !graph_disconnected {
                match graph_receiver.try_recv() {
                    Ok(graph) => {
                        while !pipe.in_flight.is_empty() {
                            finish_oldest!(); // the old network answers what it was given
                        }
                        handle_new_graph(&mut network, graph, &mut load_network, &thread_name);
                        continue;
                    }
                    Err(TryRecvError::Empty) => {}
                    Err(TryRecvError::Disconnected) => graph_disconnected = true, // keep the final network (:119-143)
                }
            }
            if !jobs_disconnected {
                drain_jobs!();
            }
            if ready(&state, &pipe, jobs_disconnected) && pipe.in_flight.len() < depth {
                submit_one!();
            } else {
                finish_oldest!();
            }
            continue;
        }

        if jobs_disconnected {
            // the job channel has disconnected: evaluate what is left, then exit (executor.rs:101-115)
            while state.x.len() > 0 {
                submit_one!();
                finish_oldest!();
            }
            assert!(state.items_to_eval() == 0 && state.items_to_send() == 0);
            return;
        }

        // nothing in flight: block exactly like the synchronous loop (executor.rs:48-66)
        let mut selector = Selector::new();
        if !graph_disconnected {
            selector = selector.recv(&graph_receiver, Message::Graph);
        }
        if network.is_some() {
            selector = selector.recv(&job_receiver, Message::Job);
        }
        begin_event_with_color("wait", CL_YELLOW);
        let message = selector.wait();
        end_event();

        match message {
            Message::Graph(Ok(graph)) => handle_new_graph(&mut network, graph, &mut load_network, &thread_name),
            Message::Graph(Err(RecvError::Disconnected)) => match &network {
                Some(_) => graph_disconnected = true,
                None => {
                    assert!(state.items_to_eval() == 0, "Executor {}: graph disconnected with items pending", thread_name);
                    match job_receiver.recv() {
                        Ok(_) => panic!("Executor {}: got new job after graph disconnection", thread_name),
                        Err(RecvError::Disconnected) => {}
                    }
                    return;
                }
            },
            Message::Job(Ok(job)) => {
                pipe.note_pushed(job.x.len());
                state.push_job_pipelined(job);
                drain_jobs!();
                if !jobs_disconnected && ready(&state, &pipe, false) {
                    submit_one!();
                }
            }
            Message::Job(Err(RecvError::Disconnected)) => jobs_disconnected = true,
        }
    }
}

impl<X, Y> State<X, Y> {
    /// push_job (executor.rs:222-238) without the invariant check, which does not know about in-flight inputs
    fn push_job_pipelined(&mut self, job: Job<X, Y>) {
        let Job { x, sender } = job;
        if x.len() == 0 {
            let _ = sender.send(vec![]);
        } else {
            self.senders.push_back((x.len(), sender));
            self.x.extend(x.into_iter());
        }
    }
}

/// respond_batch (executor.rs:276-301) for results whose inputs already left `state.x` at submit time
fn respond_in_flight<X, Y>(state: &mut State<X, Y>, batch_y: Vec<Y>) {
    let batch_size = batch_y.len();
    if state.leftover_y.is_empty() && state.senders[0].0 == batch_size {
        let _ = state.senders.pop_front().unwrap().1.send(batch_y);
    } else {
        state.leftover_y.extend(batch_y.into_iter());
        while state.can_fill_next_sender() {
            let (count, sender) = state.senders.pop_front().unwrap();
            let block_y = state.leftover_y.drain(0..count).collect_vec();
            let _ = sender.send(block_y);
        }
    }
}

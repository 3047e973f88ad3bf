/* dtrl_trainer.h -- C ABI of the MI355X-native MACE trainer step (libdtrl.so): batch-32 forward / backward / Caffe-SGD update of the *_mace3 (and single-head)
 * nets as hand-written HIP GEMM kernels with weights, solver history, activations and the replay rows resident on the device.
 *
 * The reference has no FFI here either; the seam is cNeuralNetTrainer / cMACETrainer's use of cNeuralNet (learning/NeuralNetTrainer.cpp:696-784,
 * learning/MACETrainer.cpp:163-250, 346-372, 577-633; learning/NeuralNet.cpp:352-375, 1077-1122), which wraps Caffe's Forward and SGDSolver::Step.
 * Every entry point below names the reference call it replaces. Plain pointers and sizes; "dev" pointers are device memory of the trainer's GPU
 * (e.g. a framework tensor's data pointer), everything else is host memory. All calls are queued on the stream given to dtrl_trainer_set_stream
 * (default: the legacy default stream) and return without waiting unless stated; dtrl_trainer_sync waits. Status codes as in dtrl.h (0 = ok). */
#ifndef DTRL_TRAINER_H
#define DTRL_TRAINER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dtrl_trainer dtrl_trainer;

/* Topology (the deploy / train prototxt of the family: slice -> 3 x Convolution + ReLU -> terr_ip0 + ReLU -> concat with the character slice -> trunk
 * InnerProduct + ReLU -> n_heads x [InnerProduct + ReLU -> InnerProduct]) and the solver's hyper-parameters (base_lr, momentum, weight_decay of the solver
 * prototxt; lr_policy "fixed"). MACE nets: head 0 = the critic (n_frags values), heads 1 .. n_frags = the actors (frag_size each); outputs are laid
 * out head after head. Single-head nets (Q net, CACLA actor): n_heads = 1, n_frags = 0. Weight vector = Caffe blob order, weight then bias per layer. */
typedef struct dtrl_trainer_desc {
	int32_t state_size, n_terrain;
	int32_t conv_ch[3], conv_k[3];
	int32_t fc_terr, fc_trunk, fc_head;
	int32_t n_heads, head_out[8];
	int32_t n_frags, frag_size;
	int32_t batch, max_eval;          /* solver batch (MemoryData batch_size, 32) and the largest evaluation batch (>= 2 x batch for dtrl_trainer_actor_filter) */
	float base_lr, momentum, weight_decay, discount;
	int32_t freeze_target;            /* cMACETrainer::EnableTargetNet(): != 0 -> critic targets come from the frozen copy (dtrl_trainer_update_target), else from the current net */
} dtrl_trainer_desc;

/* Replaces: cNeuralNet::LoadNet + LoadSolver (learning/NeuralNet.cpp:53-108): builds the net and the solver state on device `device_id` (-1: current). */
int dtrl_trainer_create(const dtrl_trainer_desc* desc, int device_id, dtrl_trainer** out);
void dtrl_trainer_destroy(dtrl_trainer* t);
const char* dtrl_trainer_last_error(const dtrl_trainer* t);
/* every later call is queued on this HIP stream (hipStream_t as a pointer; NULL = the legacy default stream), so it is ordered with the caller's own work */
int dtrl_trainer_set_stream(dtrl_trainer* t, void* hip_stream);
int dtrl_trainer_sync(dtrl_trainer* t);

int64_t dtrl_trainer_num_params(const dtrl_trainer* t);
/* Replaces: cNeuralNet::CopyModel / LoadModel / GetParams (learning/NeuralNet.cpp:636-658): which = 0 current net, 1 target net, 2 solver history,
 * 3 lr_mult per element, 4 decay_mult per element (the per-blob multipliers of the train prototxt, expanded). Host arrays of num_params floats; synchronous. */
int dtrl_trainer_set_params(dtrl_trainer* t, int which, const float* host, int64_t n);
int dtrl_trainer_get_params(dtrl_trainer* t, int which, float* host, int64_t n);
/* test / diagnosis: the first n floats of an internal device buffer (10 eval input, 11 eval output, 12 new_q, 13 train input, 14 train output,
 * 15 loss gradient, 16 weight gradient); what cNeuralNet::GetLayerState offers for blobs. Synchronous. */
int dtrl_trainer_debug_get(dtrl_trainer* t, int which, float* host, int64_t n);
/* device pointer of the flat weight vector (which = 0 / 1): hand it to dtrl_set_policy_device without a host round trip (cNeuralNetLearner::SyncNet) */
int dtrl_trainer_params_device(dtrl_trainer* t, int which, float** dev_ptr);
/* Replaces: cNeuralNet::SetInputOffsetScale / SetOutputOffsetScale (learning/NeuralNet.cpp:217-260). Host doubles, state_size / out_size entries; NULL keeps. */
int dtrl_trainer_set_normalizers(dtrl_trainer* t, const double* in_off, const double* in_scale, const double* out_off, const double* out_scale);
/* Replaces: cMACETrainer::UpdateTargetNet (learning/MACETrainer.cpp:515-573): target <- current. */
int dtrl_trainer_update_target(dtrl_trainer* t);

/* Replaces: cNeuralNet::EvalBatch (learning/NeuralNet.cpp:377-410): Y_dev[n][out] = un-normalised outputs of net `which` on X_dev[n][state_size], n <= max_eval. */
int dtrl_trainer_eval(dtrl_trainer* t, int which, const float* X_dev, int n, float* Y_dev);
/* Replaces: cNeuralNet::Train (learning/NeuralNet.cpp:1077-1122) on one batch: data and labels normalised, EuclideanLoss, one Caffe SGD iteration
 * (L2 regularisation with weight_decay x decay_mult, history = momentum x history + base_lr x lr_mult x diff, w -= history). X_dev[batch][state_size],
 * Y_dev[batch][out] un-normalised labels. The loss lands in dtrl_trainer_loss()[0]. */
int dtrl_trainer_step(dtrl_trainer* t, const float* X_dev, const float* Y_dev);

/* The replay memory the MACE calls below read: rows [mem_size][W] float32 in the MACE layout [r | s | a = (fragment id, params) | s'] exactly as
 * dtrl_drain_tuples emits them, and one int64 flag word per row (bit 0 = fail). Device pointers that stay valid for the trainer's life (mPlaybackMem).
 * A single-head net (Q, CACLA critic / actor: n_frags == 0) binds its replay memory [r | s | a | s'] of any action width for the staged stores alone. */
int dtrl_trainer_bind_replay(dtrl_trainer* t, const float* mem_dev, const int64_t* flags_dev, int W);
/* New tuples without a queued copy or a framework call. dtrl_trainer_stage_rows / _flags: page-locked, device-visible staging arrays owned by the trainer
 * ([dtrl_trainer_stage_capacity()][W] float32 rows as dtrl_drain_tuples emits them, one int64 flag word per row; valid after dtrl_trainer_bind_replay, NULL before).
 * dtrl_trainer_add_staged moves staged rows [first, first + n) into the replay slots (head + i) % mem_size on the trainer's stream, ordered with the steps
 * queued around it. Replaces: cNeuralNetTrainer::AddTuple's SetTuple(mBufferHead, tuple) (learning/NeuralNetTrainer.cpp:145-165); CheckTuple, the head and the
 * actor / critic index buffers stay with the caller. The staged rows must stay untouched until the call has executed (dtrl_trainer_sync, or any later call
 * whose result the host has read). */
float* dtrl_trainer_stage_rows(dtrl_trainer* t);
int64_t* dtrl_trainer_stage_flags(dtrl_trainer* t);
int dtrl_trainer_stage_capacity(dtrl_trainer* t);
int dtrl_trainer_add_staged(dtrl_trainer* t, int first, int n, int64_t head, int64_t mem_size);
/* page-locked, device-visible host arrays owned by the trainer: the caller writes replay slots into idx ([0, batch): critic batch; [batch, 2 batch): actor
 * candidates; [max_eval, max_eval + batch): actor batch -- separate windows, so that queued work never sees a later call's indices) and reads `better` and `loss` after dtrl_trainer_sync -- no copy is queued in either direction */
int64_t* dtrl_trainer_idx(dtrl_trainer* t);
int32_t* dtrl_trainer_better(dtrl_trainer* t);
float* dtrl_trainer_loss(dtrl_trainer* t);
/* Replaces: cMACETrainer::BuildProblemX / BuildProblemY (new_q = r (1 - discount) [+ discount max_f Q_target(s')] written over the taken fragment's value,
 * learning/MACETrainer.cpp:163-250, 478-515) + cNeuralNet::Train on the batch idx[0 .. batch). loss -> dtrl_trainer_loss()[0]. */
int dtrl_trainer_critic_step(dtrl_trainer* t);
/* Replaces: the test of cMACETrainer::UpdateActorBatchBuffer (learning/MACETrainer.cpp:577-609) on the candidates idx[batch .. batch + n), n <= batch, 2 n <= max_eval:
 * better[m] = (new_q(tuple m) > max_f Q_target(s_m)[f]). */
int dtrl_trainer_actor_filter(dtrl_trainer* t, int n);
/* dtrl_trainer_critic_step and dtrl_trainer_actor_filter(batch) in one pass, for a FROZEN target net (freeze_target != 0, max_eval >= 3 x batch): Q_target is
 * needed on s' of the critic batch and on s, s' of the actor candidates, and none of them depends on the critic update, so the target net is evaluated once
 * over all three. idx[batch .. 2 batch) must hold `batch` valid slots (pad a shorter candidate list by repeating one; ignore the padded answers). */
int dtrl_trainer_critic_step_and_filter(dtrl_trainer* t);
/* Replaces: cMACETrainer::BuildActorProblemY + StepActor (learning/MACETrainer.cpp:285-305, 611-633) on idx[max_eval .. max_eval + batch): labels = the net's
 * own outputs with the taken fragment's parameters replaced by the tuple's action. loss -> dtrl_trainer_loss()[1]. */
int dtrl_trainer_actor_step(dtrl_trainer* t);


/* ---- for a C++ host that speaks the reference's cNeuralNet interface (include/BatchNeuralNet.h): files in, host arrays in and out ----
 * Replaces cNeuralNet::LoadNet + LoadSolver (learning/NeuralNet.cpp:62-79, 110-136) WITH the file reading: net_file = the deploy or train prototxt of the family,
 * solver_file = the solver prototxt (NULL / "": evaluation only); the train net named by the solver (`net: "..."`, resolved against data_root, else
 * <x>_solver -> <x>_train, else <net>_deploy -> <net>_train) supplies the MemoryData batch_size and the per-blob lr_mult / decay_mult. max_eval = 3 x batch. */
int dtrl_trainer_create_from_files(const char* net_file, const char* solver_file, const char* data_root, float discount, int freeze_target, int device_id, dtrl_trainer** out);
int dtrl_trainer_dims(const dtrl_trainer* t, int* in_size, int* out_size, int* batch, int* max_eval);
/* what a freshly built Caffe net holds: "xavier" weights (uniform +- sqrt(3 / fan_in)), zero biases, in both nets; history cleared */
int dtrl_trainer_init_xavier(dtrl_trainer* t, uint64_t seed);
/* cNeuralNet::Eval / EvalBatch (learning/NeuralNet.cpp:352-387) and cNeuralNet::Train (:229-245) on HOST arrays of doubles (Eigen's storage, row-major copies):
 * staged through device buffers of the trainer, synchronous. Any n for the evaluation; exactly `batch` rows for the step. */
int dtrl_trainer_eval_host(dtrl_trainer* t, int which, const double* X, int n, double* Y);
int dtrl_trainer_step_host(dtrl_trainer* t, const double* X, const double* Y, double* loss);
int dtrl_trainer_get_normalizers(dtrl_trainer* t, double* in_off, double* in_scale, double* out_off, double* out_scale);
/* cNeuralNet::CopyModel (learning/NeuralNet.cpp:722-733): parameters + normalisers of src's current net into dst's, device to device */
int dtrl_trainer_copy_model(dtrl_trainer* dst, dtrl_trainer* src);

/* ---- data-parallel step: every rank steps on a minibatch of ITS OWN tuples, one all-reduce of the flat gradient, the identical update on all ranks ----
 * Stands in for the reference's answer to trainer fan-in -- a pool of learners pushing gradients to cParamServer::UpdateNet (learning/ParamServer.cpp:65-90,
 * learning/AsyncMACETrainer.cpp:14-45; asynchronous, out of scope as code) -- in its synchronous form (last row: 570 474 floats = 2.28 MB per all-reduce).
 * The gradient buffer holds num_params + 1 floats: the mean gradient over the rank's `batch` rows and, behind it, the number of rows (batch, or 0 after
 * dtrl_trainer_zero_grad). The caller SUMS it over the ranks (RCCL all-reduce on the trainer's stream); dtrl_trainer_apply_grad rescales by batch / total rows, so
 * ranks that had no batch this round simply do not count, and applies the Caffe SGD rule. With one rank and no all-reduce, grad + apply == the fused step. */
/* the caller's device buffer of num_params + 1 floats becomes the gradient buffer (e.g. a framework tensor a collective can take); NULL = the trainer's own. Synchronises. */
int dtrl_trainer_bind_grad(dtrl_trainer* t, float* grad_dev);
int dtrl_trainer_grad_device(dtrl_trainer* t, float** grad_dev);
/* dtrl_trainer_step / dtrl_trainer_critic_step / dtrl_trainer_actor_step up to and including the backward pass: the gradient is left in the buffer, nothing is updated
 * (cNeuralNet::ForwardBackward, learning/NeuralNet.cpp:247-261, is the reference's call of this shape: cNeuralNetTrainer::UpdateNet's asynchronous branch) */
int dtrl_trainer_grad_step(dtrl_trainer* t, const float* X_dev, const float* Y_dev);
int dtrl_trainer_critic_grad(dtrl_trainer* t);
int dtrl_trainer_actor_grad(dtrl_trainer* t);
int dtrl_trainer_zero_grad(dtrl_trainer* t);
/* Caffe SGD update from the (summed) gradient buffer; the total row count lands in dtrl_trainer_loss()[slot], slot = 2 (critic round) or 3 (actor round); count 0 = no update */
int dtrl_trainer_apply_grad(dtrl_trainer* t, int slot);

/* ---- single-head trainers (n_frags == 0) on replay rows [r | s | a (A entries) | s']: one recorded launch sequence per iteration, no framework op between the
 * minibatch's slots and the updated weights ----
 * dtrl_trainer_value_step: slots dtrl_trainer_idx()[0 .. batch); loss -> dtrl_trainer_loss()[0].
 *   kind 0 = cQNetTrainer's solver iteration (learning/QNetTrainer.cpp:27-83, 142-163): new_q = r (1 - discount) [+ discount max_a' Q(s')[a'] unless the tuple failed], the
 *            net itself as reference net (pool of one); label = the net's own outputs, the entry of the action taken (first maximum of the one-hot block) := new_q. A == out_size.
 *   kind 1 = cCaclaTrainer's critic iteration (learning/CaclaTrainer.cpp:149-157, 234-277): new_v = r (1 - discount) [+ discount V_target(s')]; label = new_v. out_size == 1. */
int dtrl_trainer_value_step(dtrl_trainer* t, int kind);
/* cCaclaTrainer::UpdateActorBatchBuffer's TD test (learning/CaclaTrainer.cpp:342-387) on the critic's trainer: candidates dtrl_trainer_idx()[batch .. batch + n), n <= batch;
 * after dtrl_trainer_sync: dtrl_trainer_td()[m] = new_v(s'_m) - V_target(s_m), dtrl_trainer_better()[m] = (td > 0) */
int dtrl_trainer_td_filter(dtrl_trainer* t, int n);
float* dtrl_trainer_td(dtrl_trainer* t);
/* cCaclaTrainer's actor iteration (BuildTupleActorY): one solver step of THIS net towards the action parameters of the tuples in slots dtrl_trainer_idx()[max_eval ..
 * max_eval + batch) of the replay memory it is bound to (bind the actor's trainer to the critic's memory); A == out_size; loss -> dtrl_trainer_loss()[0] */
int dtrl_trainer_action_step(dtrl_trainer* t);

#ifdef __cplusplus
}
#endif
#endif

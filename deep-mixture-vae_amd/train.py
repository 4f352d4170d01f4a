"""Training CLI of the DMVAE drop-in -- same flags and flow as code/train.py of
the reference (argparse :28-97, main :101-338) for --model dmvae, with the
per-batch path on MI355X.  Run from this directory:  python train.py [flags]
Multi-GPU (one process per GPU, RCCL gradient all-reduce):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --batch_size 32768
(with more than one rank only full global batches are trained: the batch must not exceed the row count)
Flags that the reference accepts but that address models outside the DMVAE hot
path (--model dmoe/dvmoe/vademoe, --visdom) are parsed and rejected /
ignored with a message, see SURVEY.md 2.1.  --plotting writes the reference's
two figures (regenerated.png, sampled.png) as PNG grids.  --pretrain runs the two
pretraining stages of base_models.py:304-423 (recon-only Adam at epsilon = 0,
GMM-initialised prior tables, latent-loss Adam over the c-head).
New flags (defaults = reference behaviour): --batch_size, --dtype, --seed,
--host_noise, --gumbel, --temperature, --enc_layers, --head_dim, --dec_layers.
"""
import argparse
import json
import math
import os
import sys
import zipfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

parser = argparse.ArgumentParser(description="DMVAE training on MI355X (flag names and defaults of the reference's train.py:28-97)")

# (flag, type, default, what it does here).  Names and defaults are the reference's CLI contract; the
# descriptions say what THIS implementation does with each flag.
_FLAGS = [
    ("model", str, "dmvae", "model family; only dmvae and vade are built (the MoE families are rejected)"),
    ("model_name", str, "", "directory name under saved-models/<dataset>/ (default: the model family)"),
    ("dataset", str, "mnist", "mnist | fashion-mnist | synthetic (idx files under data/<dataset>/, else a synthetic stand-in)"),
    ("latent_dim", int, 10, "width D of the Gaussian latent z"),
    ("output_dim", int, 1, "MoE regression target width (accepted, unused: MoE is out of scope)"),
    ("n_clusters", int, -1, "mixture components K; -1 takes the dataset's class count"),
    ("n_experts", int, 5, "MoE expert count (accepted, unused)"),
    ("n_epochs", int, 500, "training epochs"),
    ("pretrain_epochs_vae", int, 200, "epochs of the reconstruction-only pretraining stage"),
    ("pretrain_epochs_prior", int, 200, "epochs of the prior stage (also max_iter of the GMM that seeds the prior tables)"),
    ("init_lr", float, 0.002, "Adam learning rate (constant: the reference's decay is a no-op, SURVEY F3)"),
    ("decay_rate", float, 0.9, "accepted and inert, as in the reference (global_step is the literal 0)"),
    ("decay_epochs", int, 25, "accepted and inert, see --decay_rate"),
    ("pretrain_vae_lr", float, 0.0005, "Adam learning rate of the reconstruction-only stage"),
    ("pretrain_decay_rate", float, 0.9, "accepted and inert"),
    ("pretrain_decay_epochs", int, 25, "accepted and inert"),
    ("pretrain_prior_lr", float, 0.0005, "Adam learning rate of the prior stage"),
    ("anneal_step", float, 0.1, "increment of the KL weight at each annealing point"),
    ("anneal_epochs", int, 1000, "epochs between two increments of the KL weight"),
    ("plot_epochs", int, 100, "epochs between two sets of sample / reconstruction grids"),
    ("save_epochs", int, 10, "epochs between two --debug stops"),
]
_SWITCHES = [
    ("classification", "MoE objective switch (accepted, unused)"),
    ("pretrain", "run the two pretraining stages before training"),
    ("kl_annealing", "start the KL weight at 0 and raise it by --anneal_step every --anneal_epochs"),
    ("plotting", "write sampled.png / regenerated.png grids every --plot_epochs"),
    ("debug", "stop in pdb on the first batch every --save_epochs"),
    ("visdom", "accepted; no Visdom server is contacted"),
    ("featLearn", "MoE feature-learning switch (accepted, unused)"),
]
for _name, _type, _default, _help in _FLAGS:
    parser.add_argument("--" + _name, type=_type, default=_default, help=_help)
for _name, _help in _SWITCHES:
    parser.add_argument("--" + _name, action="store_true", default=False, help=_help)
# ---- extensions (defaults reproduce the reference)
parser.add_argument("--batch_size", type=int, default=100, help="GLOBAL batch size (reference hard-codes 100, train.py:215-216)")
parser.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "fp32"], help="bf16 MFMA (throughput) or exact fp32 (parity)")
parser.add_argument("--seed", type=int, default=0)
parser.add_argument("--host_noise", action="store_true", default=False,
                    help="draw epsilon / gumbel on the host from NumPy like the reference instead of on-device Philox")
parser.add_argument("--gumbel", action="store_true", default=False, help="Gumbel-Softmax relaxed KL (report's variant; off = checked-in graph)")
parser.add_argument("--temperature", type=float, default=1.0)
parser.add_argument("--enc_layers", type=str, default="500,500")
parser.add_argument("--head_dim", type=int, default=2000)
parser.add_argument("--dec_layers", type=str, default="2000,500,500")
parser.add_argument("--cnn", action="store_true", default=False,
                    help="the checked-in convolutional encoder trunk (base_models.py:156,176-216) instead of the MLP branch")


def main(argv):
    import torch
    import base_models
    from includes.utils import load_data, Dataset
    from includes import visualization
    from dmvae_hip import Session

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    np.random.seed(argv.seed)

    model_str, model_name = argv.model, argv.model_name
    if model_str[-3:] == "moe":
        raise NotImplementedError("--model %s: only the ELBO path of the clustering VAEs (dmvae, vade) is built (SURVEY.md 2.1, 8)" % model_str)
    if model_str not in ("dmvae", "vade"):
        raise NotImplementedError
    plotting = argv.plotting and rank == 0 and argv.dataset == "mnist"   # train.py:157-163: plots exist for the image sets
    if argv.visdom and rank == 0:
        print("--visdom: accepted, not used")

    dataset = load_data(argv.dataset, classification=argv.classification, output_dim=argv.output_dim)
    if rank == 0:
        print(dataset.input_type)
        if dataset.synthetic:
            print("no idx files under data/%s: training on the deterministic synthetic stand-in" % dataset.datagroup)
    if model_name == "":
        model_name = model_str
    n_clusters = argv.n_clusters
    if n_clusters < 1:
        n_clusters = dataset.n_classes
    if argv.batch_size % world:
        raise ValueError("--batch_size must be divisible by the number of ranks")

    sess = Session()
    if model_str == "vade":           # train.py:199-203; its layer widths are literals of the class (2000,500,500 / 500,500,2000)
        model = base_models.VaDE(
            model_name, dataset.input_type, dataset.input_dim, argv.latent_dim, n_clusters,
            activation="relu", initializer="xavier", cnn=argv.cnn,
            batch_size=argv.batch_size // world, dtype=argv.dtype,
            noise="host" if argv.host_noise else "device", seed=argv.seed, session=sess
        ).build_graph()
    else:
        model = base_models.DeepMixtureVAE(
            model_name, dataset.input_type, dataset.input_dim, argv.latent_dim, n_clusters,
            activation="relu", initializer="xavier", cnn=argv.cnn,
            batch_size=argv.batch_size // world, dtype=argv.dtype,
            enc_layers=[int(v) for v in argv.enc_layers.split(",")], head_dim=argv.head_dim,
            dec_layers=[int(v) for v in argv.dec_layers.split(",")], gumbel=argv.gumbel, temperature=argv.temperature,
            noise="host" if argv.host_noise else "device", seed=argv.seed, session=sess
        ).build_graph()

    # dmvae trains on train + test rows (train.py:205-213)
    train_data = np.concatenate([dataset.train_data, dataset.test_data], axis=0)
    train_classes = np.concatenate([dataset.train_classes, dataset.test_classes], axis=0)
    test_data = Dataset((dataset.test_data, dataset.test_classes), batch_size=argv.batch_size)
    train_data = Dataset((train_data, train_classes), batch_size=argv.batch_size)

    model.define_train_step(argv.init_lr, train_data.epoch_len * argv.decay_epochs, argv.decay_rate)

    model.path = "saved-models/%s/%s" % (dataset.datagroup, model.name)
    if rank == 0:
        for path in [model.path + "/" + x for x in ["model", "vae", "prior"]]:
            if not os.path.exists(path):
                os.makedirs(path)
    if argv.pretrain:     # train.py:222-231, :241-249
        model.define_pretrain_step(argv.pretrain_vae_lr, argv.pretrain_prior_lr)
        model.pretrain(sess, train_data, argv.pretrain_epochs_vae, argv.pretrain_epochs_prior)
    ckpt_path = model.path + "/model/parameters.ckpt"
    try:      # the checkpoint is an npz archive of the trainables (as the pretraining stages write): no pickle
        with np.load(ckpt_path, allow_pickle=False) as f:
            model.load_state_dict({k: f[k] for k in f.files})
        if rank == 0:
            print("Restored", ckpt_path)
    except (OSError, ValueError, KeyError, RuntimeError, EOFError, zipfile.BadZipFile) as e:
        # no checkpoint, one written by another revision (e.g. a torch.save zip at this path), a truncated archive,
        # other shapes / names: say why and carry on from the initial parameters, as the reference does
        # (bare except around saver.restore, train.py:251-258) and as DeepMixtureVAE._restore does
        if rank == 0:
            print("Could not load trained model" + ("" if isinstance(e, FileNotFoundError) else " (%s: %s)" % (type(e).__name__, e)))
    if world > 1:   # every rank starts from rank 0's parameters
        import torch.distributed as dist
        dist.broadcast(model.engine.param, src=0)
        model.engine.refresh_shadow()

    from tqdm import tqdm
    maxAcc = 0.0
    with tqdm(range(argv.n_epochs), postfix={"loss": "inf", "accTrain": "0.00%", "accTest": "0.00%"}, disable=rank != 0) as bar:
        anneal_term = 0.0 if argv.kl_annealing else 1.0
        for epoch in bar:
            if epoch % argv.save_epochs == 0 and argv.debug:
                model.debug(sess, train_data)
            if plotting and epoch % argv.plot_epochs == 0:      # train.py:281-286
                visualization.mnist_sample_plot(model, sess)
                visualization.mnist_regeneration_plot(model, test_data, sess)
            if argv.kl_annealing and (epoch + 1) % argv.anneal_epochs == 0:
                anneal_term = min(anneal_term + argv.anneal_step, 1.0)
            loss = model.train_op(sess, train_data, anneal_term)
            accTrain = model.get_accuracy(sess, train_data)
            accTest = model.get_accuracy(sess, test_data)
            improved = accTest > maxAcc
            if world > 1:
                # the branch below holds a COLLECTIVE (sync_master): rank 0 decides and every rank follows.  Each rank scores the same
                # test set with the same weights, but an accuracy may still differ by a sample between ranks (VaDE's k noise draws from
                # the process-global NumPy RNG; a last-bit difference in a logit) -- ranks deciding for themselves could then split
                # at the all-gather and hang the job
                import torch
                import torch.distributed as dist
                flag = torch.tensor([1 if improved else 0], dtype=torch.int32, device=model.engine.device if dist.get_backend() == "nccl" else "cpu")
                dist.broadcast(flag, src=0)
                improved = bool(int(flag.item()))
            if improved:
                maxAcc = max(maxAcc, accTest)
                if world > 1:      # sharded bf16 steps leave a rank's fp32 weights current on its own slice only
                    from dmvae_hip import make_exchange
                    model.engine.sync_master(make_exchange(4 * model.engine.param.numel()))
                if rank == 0:
                    with open(ckpt_path + ".tmp", "wb") as f:
                        np.savez(f, **model.state_dict())
                    os.replace(ckpt_path + ".tmp", ckpt_path)
            if rank == 0:      # one JSON line per epoch (the reference has no metrics log; its tqdm postfix is the only record)
                ep = dict(getattr(model, "last_epoch", None) or {})
                sec = ep.pop("seconds", None)
                rec = dict(epoch=epoch, **ep, images_per_sec=(ep.get("rows", 0) / sec if sec else None), train_seconds=sec,
                           acc_train=float(accTrain), acc_test=float(accTest), max_acc=float(max(maxAcc, accTest) if improved else maxAcc),
                           world=world, batch_size=argv.batch_size, dtype=argv.dtype, model=model_str)
                with open(argv.model + "_metrics.jsonl", "a") as fl:
                    fl.write(json.dumps(rec) + "\n")
            if math.isnan(loss):
                raise FloatingPointError("loss is NaN at epoch %d (the reference drops into pdb here, train.py:320-321)" % epoch)
            bar.set_postfix({"loss": "%.4f" % loss, "accTrain": "%.4f" % accTrain, "accTest": "%.4f" % accTest,
                             "maxAcc": "%.4f" % maxAcc, "accClusteringTest": "%.4f" % accTest})
    if plotting:                                                  # train.py:332-334
        visualization.mnist_sample_plot(model, sess)
        visualization.mnist_regeneration_plot(model, test_data, sess)
    if rank == 0:
        with open(argv.model + "_logs.txt", "a+") as fl:
            fl.write("\n" + str(argv) + "\n------\n")
            fl.write("Max Accuracy        " + str(maxAcc) + "\n============")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return loss


if __name__ == "__main__":
    args = parser.parse_args()
    print(args)
    main(args)

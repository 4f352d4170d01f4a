"""Training CLI of the DMVAE drop-in -- same flags and flow as code/train.py of
the reference (argparse :28-97, main :101-338) for --model dmvae, with the
per-batch path on MI355X.  Run from this directory:  python train.py [flags]
Multi-GPU (one process per GPU, RCCL gradient all-reduce):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --batch_size 65536
Flags that the reference accepts but that address models outside the DMVAE hot
path (--model vade/dmoe/dvmoe/vademoe, --visdom) are parsed and rejected /
ignored with a message, see SURVEY.md 2.1.  --plotting writes the reference's
two figures (regenerated.png, sampled.png) as PNG grids.  --pretrain runs the two
pretraining stages of base_models.py:304-423 (recon-only Adam at epsilon = 0,
GMM-initialised prior tables, latent-loss Adam over the c-head).
New flags (defaults = reference behaviour): --batch_size, --dtype, --seed,
--host_noise, --gumbel, --temperature, --enc_layers, --head_dim, --dec_layers.
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

parser = argparse.ArgumentParser(description="Training file for DMVAE and DVMOE")

parser.add_argument("--model", type=str, default="dmvae", help="Model to use [dmvae, vade, dmoe, dvmoe, vademoe]")
parser.add_argument("--model_name", type=str, default="", help="Name of the model")
parser.add_argument("--dataset", type=str, default="mnist", help="Dataset to use [mnist, spiral, cifar10]")
parser.add_argument("--latent_dim", type=int, default=10, help="Number of dimensions for latent variable Z")
parser.add_argument("--output_dim", type=int, default=1, help="Output dimension for regression variable for ME models")
parser.add_argument("--n_clusters", type=int, default=-1, help="Number of clusters to use")
parser.add_argument("--n_experts", type=int, default=5, help="Number of experts to use for MoE models")
parser.add_argument("--classification", action="store_true", default=False,
                    help="Whether the objective is classification or regression (ME models)")
parser.add_argument("--n_epochs", type=int, default=500, help="Number of epochs for training the model")
parser.add_argument("--pretrain_epochs_vae", type=int, default=200, help="Number of epochs for pretraining the vae model")
parser.add_argument("--pretrain_epochs_prior", type=int, default=200, help="Number of epochs for pretraining the gmm model")
parser.add_argument("--init_lr", type=float, default=0.002, help="Initial learning rate for training")
parser.add_argument("--decay_rate", type=float, default=0.9,
                    help="Decay rate for exponentially decaying learning rate (< 1.0)")
parser.add_argument("--decay_epochs", type=int, default=25,
                    help="Number of epochs between exponentially decay of learning rate")
parser.add_argument("--pretrain", action="store_true", default=False, help="Whether to pretrain the model or not")
parser.add_argument("--pretrain_vae_lr", type=float, default=0.0005, help="Initial learning rate for pretraining the vae")
parser.add_argument("--pretrain_decay_rate", type=float, default=0.9,
                    help="Decay rate for exponentially decaying learning rate (< 1.0) for pretraining")
parser.add_argument("--pretrain_decay_epochs", type=int, default=25,
                    help="Number of epochs between exponentially decay of learning rate for pretraining")
parser.add_argument("--pretrain_prior_lr", type=float, default=0.0005, help="Initial learning rate for pretraining the prior")
parser.add_argument("--kl_annealing", action="store_true", default=False,
                    help="Whether to anneal the KL term while training or not")
parser.add_argument("--anneal_step", type=float, default=0.1, help="Step size for annealing")
parser.add_argument("--anneal_epochs", type=int, default=1000, help="Number of epochs before annealing the KL term")
parser.add_argument("--plotting", action="store_true", default=False,
                    help="Whether to generate sampling and regeneration plots")
parser.add_argument("--plot_epochs", type=int, default=100, help="Nummber of epochs before generating plots")
parser.add_argument("--save_epochs", type=int, default=10, help="Nummber of epochs before saving model")
parser.add_argument("--debug", action="store_true", default=False, help="Whether to debug the models or not")
parser.add_argument("--visdom", action="store_true", default=False, help="Using visdom for plotting")
parser.add_argument("--featLearn", action="store_true", default=False, help="Whether to use feature learning in MOE")
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
    if model_str[-3:] == "moe" or model_str == "vade":
        raise NotImplementedError("--model %s: only the DMVAE ELBO path is built (SURVEY.md 2.1, 8)" % model_str)
    if model_str != "dmvae":
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
    try:
        model.load_state_dict(torch.load(ckpt_path, weights_only=False))
        if rank == 0:
            print("Restored", ckpt_path)
    except Exception:
        if rank == 0:
            print("Could not load trained model")
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
            if accTest > maxAcc:
                maxAcc = accTest
                if rank == 0:
                    torch.save(model.state_dict(), ckpt_path)
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

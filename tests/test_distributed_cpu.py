"""World-size-2 gloo tests (CPU) of the data-parallel step harness: flat-bucket gradient all-reduce,
replica broadcast, identical Adam updates.  The model here is the CPU oracle (tests may use it); the
harness code under test (train_step.TrainStep / FlatGradBucket) is the one bench.py runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

CFG = dict(input_nc=10, output_nc=3, ngf=8, filter_type='edgeconvtransinv', norm='instance', n_blocks=2, n_levels=1,
           pooling_type='max')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import stin_oracle
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    torch.manual_seed(100 + rank)                       # DIFFERENT init per rank: broadcast must fix it
    net = stin_oracle.define_G(**CFG)
    step = TrainStep(net, lr=1e-3, amsgrad=True)
    p0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    sample = make_synthetic_mesh(300 + 100 * rank, 2, seed=rank, dilations=())     # unequal scenes
    loss = step(sample)
    flat_grad = step.bucket.flat.clone()
    p1 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.save({'p0': p0, 'p1': p1, 'grad': flat_grad, 'loss': loss}, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_gradient_allreduce_matches_manual_average(tmp_path, world):
    """SURVEY 8(e) validation at world 2 and at the node size the bench targets (8 ranks, tiny meshes): the reduced flat bucket
    on every rank == the mean of the single-rank gradients of the `world` scenes, identical replicas before and after."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(tmp_path / ('r%d.pt' % r)) for r in range(world)]
    r0 = rs[0]
    for r1 in rs[1:]:
        assert torch.equal(r0['p0'], r1['p0']), 'replicas identical after the flat broadcast'
        assert torch.equal(r0['grad'], r1['grad']), 'all-reduced gradients identical on every rank'
        assert torch.equal(r0['p1'], r1['p1']), 'identical Adam update on every rank'
    assert not torch.equal(r0['p0'], r0['p1'])
    # single-process reference: mean of the two per-scene gradients at the broadcast weights
    from oracle import stin_oracle
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    from surface_texture_inpainting_net_amd.train_step import compute_loss, graph_forward
    torch.manual_seed(100)
    net = stin_oracle.define_G(**CFG)
    torch.nn.utils.vector_to_parameters(r0['p0'], net.parameters())
    grads = []
    for rank in range(world):
        net.zero_grad(set_to_none=True)
        s = make_synthetic_mesh(300 + 100 * rank, 2, seed=rank, dilations=())
        compute_loss(graph_forward(net, s), s.color, s.mask).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
    want = torch.stack(grads).double().mean(0)
    assert float((r0['grad'].double() - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9


def test_flat_bucket_views_survive_zero_grad():
    from surface_texture_inpainting_net_amd.train_step import FlatGradBucket
    lin = torch.nn.Linear(4, 3)
    b = FlatGradBucket(lin.parameters())
    assert b.flat.numel() == 15
    lin(torch.ones(2, 4)).sum().backward()
    assert float(b.flat.abs().sum()) > 0 and lin.weight.grad.data_ptr() == b.flat.data_ptr()
    lin.zero_grad(set_to_none=True)
    b.zero()
    assert lin.weight.grad is not None and float(b.flat.abs().sum()) == 0.0


SCMN = dict(feature_number=10, num_propagation_steps=1, filter_sizes=[8, 12], num_classes=4)


def _xent(model, sample):
    """The segmentation trainer's objective (trainers/segmentation_trainer.py:139-148): cross entropy of the per-vertex
    class scores against data.labels."""
    return torch.nn.functional.cross_entropy(model(sample), sample.labels)


def _scmn_sample(rank):
    from surface_texture_inpainting_net_amd.synthetic import make_synthetic_mesh
    s = make_synthetic_mesh(260 + 90 * rank, 2, seed=10 + rank, dilations=())
    g = torch.Generator().manual_seed(rank)
    s['labels'] = torch.randint(0, SCMN['num_classes'], (s.x.shape[0],), generator=g)
    return s


def _scmn_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import scmn_oracle
    from surface_texture_inpainting_net_amd.train_step import TrainStep
    torch.manual_seed(7 + rank)
    net = scmn_oracle.SingleConvMeshNet(pooling_method='mean', **SCMN)
    step = TrainStep(net, lr=1e-3, amsgrad=True, loss_fn=_xent)
    p0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    loss = step(_scmn_sample(rank))
    torch.save({'p0': p0, 'p1': torch.cat([p.detach().reshape(-1) for p in net.parameters()]), 'grad': step.bucket.flat.clone(),
                'loss': loss}, os.path.join(out_dir, 's%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_with_a_loss_hook_for_singleconvmeshnet(tmp_path):
    """TrainStep(loss_fn=...): the flat-bucket data-parallel step around SingleConvMeshNet with the segmentation trainer's
    cross entropy - the reference's only multi-GPU user (torch_geometric.nn.DataParallel, segmentation_trainer.py:34-35)."""
    mp.spawn(_scmn_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 's0.pt'), torch.load(tmp_path / 's1.pt')
    assert torch.equal(r0['p0'], r1['p0']) and torch.equal(r0['grad'], r1['grad']) and torch.equal(r0['p1'], r1['p1'])
    assert not torch.equal(r0['p0'], r0['p1'])
    from oracle import scmn_oracle
    torch.manual_seed(7)
    net = scmn_oracle.SingleConvMeshNet(pooling_method='mean', **SCMN)
    torch.nn.utils.vector_to_parameters(r0['p0'], net.parameters())
    grads = []
    for rank in range(2):
        net.zero_grad(set_to_none=True)
        net.train()
        _xent(net, _scmn_sample(rank)).backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()]))
    want = (grads[0] + grads[1]) / 2
    assert float((r0['grad'] - want).abs().max()) <= 5e-6 * float(want.abs().max()) + 1e-9      # (BatchNorm: 1 vs 8 host threads)

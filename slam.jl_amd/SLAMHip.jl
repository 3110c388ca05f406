# SLAMHip.jl -- Julia 1.x binding of libslamhip.so (include/slamhip.h).
#
# Keeps the public surface of SLAM.jl's filter core (src/SLAM.jl:5-30: SlamState,
# EKFSlamState, predict, update, add_features, associate, compute_association,
# predict_observation, mpi_to_pi) and adds the in-place names ekf_predict!,
# ekf_update!, augment!.  Every numeric operation is a `ccall` into the HIP
# library; the state lives on the GPU for the life of the handle.
#
# NOTE: there is no `julia` binary in the build image, so this file is written to
# be thin and mechanical and is NOT executed by the test-suite; the same C ABI is
# exercised through the ctypes mirror in ekf.py.  The reference itself is Julia
# 0.5/0.6 syntax (`type`, `atan2`, `chol`) and does not parse on Julia >= 1.0.
module SLAMHip

export SlamState, EKFSlamState, set_state!, predict, update, add_features, associate,
       compute_association, predict_observation, mpi_to_pi,
       ekf_predict!, ekf_update!, augment!, observe!, cov_block, cov_diag, landmark_blocks, gate_mode!, gate_info, state_written!, feature_ellipses, vehicle_ellipse,
       PFSlamState, set_pose!, init_landmarks!, pf_predict!, update_known!, step!, step_async!, step_async_batch!, flush!,
       resample!, mean_pose, weights, particles, peer_blob, attach_peers!, peer_selftest, detach_peers!, comm_info

const libslamhip = get(ENV, "SLAMHIP_LIB", joinpath(@__DIR__, "libslamhip.so"))

const SLAM_OK = Cint(0)
const SLAM_PF_HALTED = Cint(1)
const SLAM_E_NOTPD = Cint(-3)
const SLAM_F32 = Cint(0)
const SLAM_F64 = Cint(1)
const FORM_CHOLESKY = Cint(0)
const FORM_JOSEPH = Cint(1)

last_error() = unsafe_string(ccall((:slam_last_error, libslamhip), Cstring, ()))

function check(rc::Cint)
    rc == SLAM_OK && return nothing
    # the reference raises PosDefException from chol (src/ekf.jl:70) / BoundsError
    error("libslamhip status $(rc): $(last_error())")
end

abstract type SlamState end                       # src/common.jl:22

"""
    EKFSlamState(x, cov; max_landmarks, device=0)

Device-resident counterpart of `EKFSlamState{T}` (src/common.jl:25-28).  `T` is
`Float32` or `Float64`.  `state.x` / `state.cov` download on access; assigning them
uploads.  Capacity is fixed at construction (the reference re-allocates `P` per
new feature, src/ekf.jl:108-109).
"""
mutable struct EKFSlamState{T<:Union{Float32,Float64}} <: SlamState
    handle::Ptr{Cvoid}
    function EKFSlamState{T}(x::AbstractVector, cov::AbstractMatrix;
                             max_landmarks::Integer = max(64, length(x) - 3), device::Integer = 0) where {T}
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:slam_ekf_create, libslamhip), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint),
                    h, T === Float32 ? SLAM_F32 : SLAM_F64, max_landmarks, device))
        s = new{T}(h[])
        finalizer(s) do st
            ccall((:slam_ekf_destroy, libslamhip), Cint, (Ptr{Cvoid},), getfield(st, :handle))
        end
        set_state!(s, x, cov)
        return s
    end
end
EKFSlamState(x::AbstractVector{T}, cov::AbstractMatrix; kw...) where {T<:Union{Float32,Float64}} =
    EKFSlamState{T}(x, cov; kw...)

handle(s::EKFSlamState) = getfield(s, :handle)

function nlandmarks(s::EKFSlamState)
    n = Ref{Cint}(0)
    check(ccall((:slam_ekf_num_landmarks, libslamhip), Cint, (Ptr{Cvoid}, Ref{Cint}), handle(s), n))
    Int(n[])
end
Base.length(s::EKFSlamState) = 3 + 2 * nlandmarks(s)

function set_state!(s::EKFSlamState{T}, x::AbstractVector, cov::AbstractMatrix) where {T}
    xv = Vector{T}(x); P = Matrix{T}(cov)                     # column-major, like the C ABI
    n = length(xv)
    size(P) == (n, n) || throw(DimensionMismatch("cov must be n x n"))
    check(ccall((:slam_ekf_set_state, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint),
                handle(s), xv, P, n, n))
    s
end

function download(s::EKFSlamState{T}, which::Symbol) where {T}
    n = length(s)
    x = which === :cov ? C_NULL : Vector{T}(undef, n)
    P = which === :x ? C_NULL : Matrix{T}(undef, n, n)
    check(ccall((:slam_ekf_get_state, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint),
                handle(s), x, P, n, n))
    which === :x ? x : which === :cov ? P : (x, P)
end

function Base.getproperty(s::EKFSlamState, f::Symbol)
    f === :x && return download(s, :x)
    f === :cov && return download(s, :cov)
    getfield(s, f)
end

function Base.setproperty!(s::EKFSlamState, f::Symbol, v)
    # `state.x, state.cov = predict(state, ...)` (sim/ekfslam-sim.jl:100) hands back `nothing`
    # placeholders from the in-place kernels: nothing to upload.
    v === nothing && return v
    if f === :x
        length(v) == length(s) || error("x and cov change size together: use set_state!(state, x, cov)")
        set_state!(s, v, download(s, :cov))
        return v
    elseif f === :cov
        size(v, 1) == length(s) || error("x and cov change size together: use set_state!(state, x, cov)")
        set_state!(s, download(s, :x), v)
        return v
    end
    setfield!(s, f, v)
end

colmajor4(M::AbstractMatrix) = Float64[M[1, 1], M[2, 1], M[1, 2], M[2, 2]]
pairs64(z::AbstractMatrix) = Matrix{Float64}(z)              # 2 x nz column-major == (range, bearing) pairs

"Single conditional wrap, src/common.jl:102-110."
function mpi_to_pi(phi::AbstractFloat)
    phi > pi && return phi - 2pi
    phi < -pi && return phi + 2pi
    phi
end

# ---- in-place operations ---------------------------------------------------------------
"predict (src/ekf.jl:8-43) in place on the device."
function ekf_predict!(s::EKFSlamState, v::Real, g::Real, wheelbase::Real, Q::AbstractMatrix, dt::Real)
    check(ccall((:slam_ekf_predict, libslamhip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cdouble),
                handle(s), v, g, wheelbase, colmajor4(Q), dt))
    s
end

"update (src/ekf.jl:46-77) in place on the device; `form = :joseph` selects the rank-2k Joseph form."
function ekf_update!(s::EKFSlamState, z::AbstractMatrix, R::AbstractMatrix, idf; form::Symbol = :cholesky)
    m = size(z, 2)
    m == 0 && return s
    ids = Vector{Int32}(vec(collect(idf)))
    check(ccall((:slam_ekf_update, libslamhip), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int32}, Cint, Ptr{Cdouble}, Cint),
                handle(s), pairs64(z), ids, m, colmajor4(R), form === :joseph ? FORM_JOSEPH : FORM_CHOLESKY))
    s
end

"add_features (src/ekf.jl:84-122) in place on the device."
function augment!(s::EKFSlamState, z::AbstractMatrix, R::AbstractMatrix)
    nn = size(z, 2)
    nn == 0 && return s
    check(ccall((:slam_ekf_augment, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}),
                handle(s), pairs64(z), nn, colmajor4(R)))
    s
end

"""
observe!(state, z, R, gate1, gate2; form = :cholesky) -> assoc::Vector{Int32}

The observation step of sim! (sim/ekfslam-sim.jl:114-120: associate, update, add_features) as one
library call with the same results; no host round trip between the gating and the update.
assoc[i] >= 1: matched landmark, 0: dropped, -1: appended as a new feature.
"""
function observe!(s::EKFSlamState, z::AbstractMatrix, R::AbstractMatrix, gate1::Real, gate2::Real;
                  form::Symbol = :cholesky)
    nz = size(z, 2)
    assoc = zeros(Int32, nz)
    nz == 0 && return assoc
    check(ccall((:slam_ekf_observe, libslamhip), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cdouble, Cdouble, Cint, Ptr{Int32}),
                handle(s), pairs64(z), nz, colmajor4(R), gate1, gate2, form == :joseph ? 1 : 0, assoc))
    assoc
end

# ---- looking at the covariance without downloading it ------------------------------------------
"cov[r, c] for the ranges r, c (1-based, like `state.cov[r, c]`) without downloading the matrix (slam_ekf_get_block)."
function cov_block(s::EKFSlamState{T}, r::AbstractUnitRange, c::AbstractUnitRange) where {T}
    out = Matrix{T}(undef, length(r), length(c))
    check(ccall((:slam_ekf_get_block, libslamhip), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ptr{Cvoid}, Cint),
                handle(s), first(r) - 1, first(c) - 1, length(r), length(c), out, max(length(r), 1)))
    out
end

"diag(state.cov) (slam_ekf_get_diag)."
function cov_diag(s::EKFSlamState{T}) where {T}
    out = Vector{T}(undef, length(s))
    check(ccall((:slam_ekf_get_diag, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(s), out))
    out
end

"The landmarks' 2 x 2 covariance blocks, 3 x N: [P[f,f]; P[f+1,f]; P[f+1,f+1]] per landmark (slam_ekf_get_landmark_blocks)."
function landmark_blocks(s::EKFSlamState{T}) where {T}
    N = div(length(s) - 3, 2)
    out = Matrix{T}(undef, N, 3)                     # the library writes three rows of N values
    check(ccall((:slam_ekf_get_landmark_blocks, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), handle(s), out))
    permutedims(out)
end

"""
    gate_mode!(s, mode)    mode in (:auto, :sweep, :grid)

How `associate` / `observe!` search the map -- the TODO of src/data-association.jl:18-20: `:sweep` visits every landmark,
`:grid` keeps a uniform grid over the landmark means on the device and visits only the cells an observation's gate can
reach (O(candidates)), `:auto` (default) takes the grid from 16384 landmarks on.  The decisions are identical.
"""
function gate_mode!(s::EKFSlamState, mode::Symbol)
    m = mode === :auto ? 0 : mode === :sweep ? 1 : mode === :grid ? 2 : throw(ArgumentError("gate mode must be :auto, :sweep or :grid"))
    check(ccall((:slam_ekf_set_gate_mode, libslamhip), Cint, (Ptr{Cvoid}, Cint), handle(s), m))
    s
end

"slam_ekf_gate_info: (form of the last gating, cells per axis, landmarks in the grid, tail, rebuilds, queries, visited, evaluated)."
function gate_info(s::EKFSlamState)
    out = zeros(Int64, 8)
    check(ccall((:slam_ekf_gate_info, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Int64}), handle(s), out))
    (form = (nothing, :sweep, :grid)[out[1] + 1], cells_per_axis = out[2], in_grid = out[3], tail = out[4],
     rebuilds = out[5], queries = out[6], visited = out[7], evaluated = out[8])
end

"""
    state_written!(s)

After writing landmark entries of `x` / `cov` through raw device views (slam_ekf_device_ptrs): refreshes what the gating
keeps beside the matrix (packed 2 x 2 blocks, variance bound, grid of means).  Assigning `s.x` / `s.cov` needs no call.
"""
function state_written!(s::EKFSlamState)
    check(ccall((:slam_ekf_state_written, libslamhip), Cint, (Ptr{Cvoid},), handle(s)))
    s
end

"feature_ellipses(x, cov) of the browser monitor (sim/browser/wsserver.jl:72-85): 5 x N [cx; cy; rx; ry; phi], on the device."
function feature_ellipses(s::EKFSlamState)
    out = Matrix{Float64}(undef, 5, nlandmarks(s))
    check(ccall((:slam_ekf_ellipses, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), handle(s), out, C_NULL))
    out
end

"The vehicle-ellipse record of monitor() (sim/browser/wsserver.jl:60-65): [cx, cy, vehicle_phi, rx, ry, phi]."
function vehicle_ellipse(s::EKFSlamState)
    out = zeros(Float64, 6)
    check(ccall((:slam_ekf_ellipses, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), handle(s), C_NULL, out))
    out
end

# ---- the reference's function surface ------------------------------------------------------
# The reference returns (x, P) and its only caller assigns them back to the state
# (sim/ekfslam-sim.jl:100,117,120).  Here the state was already updated in place, so the
# returned pair is (nothing, nothing) and `setproperty!` ignores it -- no 1.6 GB round trip.

"predict(state, vehicle, Q, dt): `vehicle` needs measured_speed, measured_gamma, wheelbase (src/ekf.jl:14-16)."
function predict(s::EKFSlamState, vehicle, Q::AbstractMatrix, dt::AbstractFloat)
    ekf_predict!(s, vehicle.measured_speed, vehicle.measured_gamma, vehicle.wheelbase, Q, dt)
    nothing, nothing
end

function update(s::EKFSlamState, z, R, idf)
    ekf_update!(s, z, R, idf)
    nothing, nothing
end

function add_features(s::EKFSlamState, z, R)
    augment!(s, z, R)
    nothing, nothing
end

"associate(state, z, R, gate1, gate2) -> (zf 2 x nf, idf 1 x nf Int, zn 2 x nn)  (src/data-association.jl:1-51)."
function associate(s::EKFSlamState, z::AbstractMatrix, R::AbstractMatrix, gate1::Real, gate2::Real)
    nz = size(z, 2)
    assoc = zeros(Int32, nz)
    if nz > 0
        check(ccall((:slam_ekf_associate, libslamhip), Cint,
                    (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Cdouble, Cdouble, Ptr{Int32}),
                    handle(s), pairs64(z), nz, colmajor4(R), gate1, gate2, assoc))
    end
    hit = findall(>(0), assoc)
    new = findall(<(0), assoc)
    z[:, hit], reshape(Int.(assoc[hit]), 1, :), z[:, new]
end

"compute_association(state, z, R, idf) -> (nis, nd)  (src/data-association.jl:53-63; x, P are the state's)."
function compute_association(s::EKFSlamState, z::AbstractVector, R::AbstractMatrix, idf::Integer)
    out = zeros(Float64, 2)
    check(ccall((:slam_ekf_nis, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
                handle(s), Float64[z[1], z[2]], idf, colmajor4(R), out))
    out[1], out[2]
end

"predict_observation(state, idf) -> (z, H) with H dense 2 x n (src/common.jl:139-165)."
function predict_observation(s::EKFSlamState, idf::Integer)
    zp = zeros(2); Hv = zeros(2, 3); Hf = zeros(2, 2)
    check(ccall((:slam_ekf_predict_observation, libslamhip), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), handle(s), idf, zp, Hv, Hf))
    H = zeros(2, length(s))
    H[:, 1:3] = Hv
    fpos = 3 + 2 * idf - 1
    H[:, fpos:fpos+1] = Hf
    zp, H
end

# ---- PFSlamState: the FastSLAM-1.0 particle filter -----------------------------------------------
# The reference declares `Particle` / `PFSlamState` (src/common.jl:14-20,31-34) and no filter code (README.md:6);
# the algorithm is the one specified from the reference's EKF building blocks (include/slamhip.h, "FastSLAM").
# Every call is one `ccall`.  A filter SHARDED over several GPUs (one Julia process per GPU, or one task per GPU in
# one process) needs no collective library either: each rank constructs its slice (`rank`, `world`), the ranks swap
# their `peer_blob`s once (MPI.Allgather, a shared file, Distributed -- 1 KB per rank) and `attach_peers!`; from then on
# the per-step scalars travel GPU to GPU and a resampling step stays on the device (include/slamhip.h, "peers").

const SLAM_PF_PEER_BLOB_BYTES = 4096

"""
    PFSlamState{T}(n, max_landmarks; seed = 0, device = 0, rank = 0, world = 1)

`PFSlamState{T}` (src/common.jl:31-34) with `n` particles IN ALL, device resident (structure of arrays:
`pose[3][n]`, `logw[n]`, `lm[max_landmarks][5][n]`).  `world > 1`: this process owns the global particle ids
`rank * n / world ... (rank + 1) * n / world - 1` on its GPU; see `peer_blob` / `attach_peers!`.
"""
mutable struct PFSlamState{T<:Union{Float32,Float64}} <: SlamState
    handle::Ptr{Cvoid}
    n::Int                  # particles on THIS rank
    max_landmarks::Int
    rank::Int
    world::Int
    function PFSlamState{T}(n::Integer, max_landmarks::Integer; seed::Integer = 0, device::Integer = 0, rank::Integer = 0,
                            world::Integer = 1) where {T}
        n % world == 0 || error("n must be divisible by the number of ranks")
        per = div(n, world)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:slam_pf_create, libslamhip), Cint,
                    (Ref{Ptr{Cvoid}}, Cint, Int64, Int64, Int64, Cint, Cint, UInt64),
                    h, T === Float32 ? SLAM_F32 : SLAM_F64, per, n, rank * per, max_landmarks, device, seed))
        s = new{T}(h[], per, max_landmarks, rank, world)
        finalizer(s) do st
            ccall((:slam_pf_destroy, libslamhip), Cint, (Ptr{Cvoid},), st.handle)
        end
        return s
    end
end

"This rank's peer blob (IPC handles of its buffers and inbox): every rank needs every rank's."
function peer_blob(s::PFSlamState)
    blob = zeros(UInt8, SLAM_PF_PEER_BLOB_BYTES)
    check(ccall((:slam_pf_export_peer, libslamhip), Cint, (Ptr{Cvoid}, Ptr{UInt8}), s.handle, blob))
    blob
end

"`blobs`: the `world` peer blobs in rank order.  Afterwards `step_async!` resamples the sharded filter on the device."
function attach_peers!(s::PFSlamState, blobs::AbstractVector)
    length(blobs) == s.world || error("one blob per rank")
    all = reduce(vcat, [Vector{UInt8}(b) for b in blobs])
    check(ccall((:slam_pf_attach_peers, libslamhip), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), s.handle, s.rank, s.world, all))
    s
end

"Collective: true when every attached peer's inbox write arrived within `timeout_ms` (call right after `attach_peers!`)."
function peer_selftest(s::PFSlamState, timeout_ms::Integer = 3000)
    ccall((:slam_pf_peer_selftest, libslamhip), Cint, (Ptr{Cvoid}, Cint), s.handle, timeout_ms) == 0
end

"Collective: remote records come home, the peers are detached (call on every rank before any rank lets its state go)."
function detach_peers!(s::PFSlamState)
    check(ccall((:slam_pf_detach_peers, libslamhip), Cint, (Ptr{Cvoid},), s.handle))
    s
end

"[ranks, 1 if peers are attached, SLAM_PF_HALTED returns so far, resamplings so far]"
function comm_info(s::PFSlamState)
    out = zeros(Int64, 4)
    check(ccall((:slam_pf_comm_info, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, out))
    out
end

"Every particle at `pose`, uniform weights (Particle.pose, src/common.jl:15)."
function set_pose!(s::PFSlamState, pose::AbstractVector)
    check(ccall((:slam_pf_set_pose, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.handle, Float64[pose[1], pose[2], pose[3]]))
    s
end

"Landmarks 1..size(xy, 2) known to every particle at xy (2 x nl) + N(0, jitter^2), covariance diag(var, var)."
function init_landmarks!(s::PFSlamState, xy::AbstractMatrix, var::Real, jitter::Real)
    check(ccall((:slam_pf_init_landmarks, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cdouble, Cdouble),
                s.handle, Matrix{Float64}(xy), size(xy, 2), var, jitter))
    s
end

"F1: control noise per particle (sim/sim-utils.jl:35-38) + the pose update of predict (src/ekf.jl:39-41)."
function pf_predict!(s::PFSlamState, V::Real, G::Real, wheelbase::Real, Q::AbstractMatrix, dt::Real)
    check(ccall((:slam_pf_predict, libslamhip), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cdouble),
                s.handle, V, G, wheelbase, colmajor4(Q), dt))
    s
end

"F2/F3: z (2 x m) with known 1-based landmark ids; per-landmark 2 x 2 EKF update, log-weights, first sightings."
function update_known!(s::PFSlamState, z::AbstractMatrix, ids, R::AbstractMatrix)
    m = size(z, 2)
    m == 0 && return s
    check(ccall((:slam_pf_update_known, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int32}, Cint, Ptr{Cdouble}),
                s.handle, pairs64(z), Vector{Int32}(vec(collect(ids))), m, colmajor4(R)))
    s
end

"""
    step!(state, V, G, wheelbase, Q, dt, z, ids, R; neff_frac = 0.75, proposal = false) -> (Neff, resampled)

One whole filter step, synchronously: predict (or the FastSLAM-2.0 proposal), the known-id updates, the weights, the
normalisation, and systematic resampling if Neff < neff_frac * n.
"""
function step!(s::PFSlamState, V::Real, G::Real, wheelbase::Real, Q::AbstractMatrix, dt::Real, z::AbstractMatrix, ids,
               R::AbstractMatrix; neff_frac::Real = 0.75, proposal::Bool = false)
    step_async!(s, V, G, wheelbase, Q, dt, z, ids, R; neff_frac = neff_frac, proposal = proposal)
    out = flush!(s)
    out[1], out[2] != 0
end

"The same step ENQUEUED (slam_pf_step_auto): Neff, the decision and the resampling stay on the device; `flush!` waits."
function step_async!(s::PFSlamState, V::Real, G::Real, wheelbase::Real, Q::AbstractMatrix, dt::Real, z::AbstractMatrix, ids,
                     R::AbstractMatrix; neff_frac::Real = 0.75, force::Integer = -1, proposal::Bool = false)
    check(ccall((:slam_pf_step_auto, libslamhip), Cint,
                (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Int32}, Cint, Ptr{Cdouble},
                 Cdouble, Cint, Cint),
                s.handle, V, G, wheelbase, colmajor4(Q), dt, pairs64(z), Vector{Int32}(vec(collect(ids))), size(z, 2),
                colmajor4(R), neff_frac, force, proposal ? 1 : 0))
    s
end

"""
K `step_async!` calls as ONE (slam_pf_step_auto_batch): `controls` is K x 2 (V, G per row), `obs` a vector of K pairs `(z, ids)`
(z 2 x m_k), `force` a vector of K integers (-1 the Neff rule, 0 never, 1 always).  Runs of at least four consecutive steps that
cannot resample go as one persistent launch where the filter allows it and `persistent = true` says that nothing else keeps the
device busy meanwhile; the result is the K calls' bit for bit.
"""
function step_async_batch!(s::PFSlamState, controls::AbstractMatrix, wheelbase::Real, Q::AbstractMatrix, dt::Real, obs::AbstractVector,
                           R::AbstractMatrix; neff_frac::Real = 0.75, force::AbstractVector = fill(-1, length(obs)),
                           proposal::Bool = false, persistent::Bool = false)
    K = length(obs)
    ms = Int32[size(o[1], 2) for o in obs]
    stride = max(1, maximum(ms; init = 0))
    zz = zeros(Float64, 2, stride, K)                 # (range, bearing) pairs, zstride pairs per step
    ii = zeros(Int32, stride, K)
    for k in 1:K
        zz[:, 1:ms[k], k] = obs[k][1]
        ii[1:ms[k], k] = vec(collect(obs[k][2]))
    end
    vg = Matrix{Float64}(transpose(Float64.(controls)))      # 2 x K: (V, G) per step, contiguous
    took = Ref{Cint}(0)
    check(ccall((:slam_pf_step_auto_batch, libslamhip), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Cint, Ptr{Cdouble},
                 Cdouble, Ptr{Int32}, Cint, Cint, Ref{Cint}),
                s.handle, K, vg, wheelbase, colmajor4(Q), dt, zz, ii, ms, stride, colmajor4(R), neff_frac,
                Vector{Int32}(force), proposal ? 1 : 0, persistent ? 2 : 0, took))
    s
end

"Wait for the queued steps: [Neff of the last step, 1.0 if it resampled, resamplings so far, steps so far]."
function flush!(s::PFSlamState)
    out = zeros(Float64, 4)
    check(ccall((:slam_pf_flush, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.handle, out))
    out
end

"F4: normalise, and resample if Neff < neff_frac * n.  Returns whether it resampled."
function resample!(s::PFSlamState, neff_frac::Real = 0.75)
    did = Ref{Cint}(0)
    check(ccall((:slam_pf_resample, libslamhip), Cint, (Ptr{Cvoid}, Cdouble, Ref{Cint}), s.handle, neff_frac, did))
    did[] != 0
end

"Weighted mean pose [x, y, phi]."
function mean_pose(s::PFSlamState)
    out = zeros(Float64, 3)
    check(ccall((:slam_pf_get_mean_pose, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.handle, out))
    out
end

"The particles' weights (Particle.weight, src/common.jl:19)."
function weights(s::PFSlamState)
    out = zeros(Float64, s.n)
    check(ccall((:slam_pf_get_weights, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), s.handle, out))
    out
end

"Download: (pose 3 x n ... as n x 3, logw n, lm n x 5 x max_landmarks): the device's SoA arrays, particle index fastest."
function particles(s::PFSlamState{T}) where {T}
    pose = Matrix{T}(undef, s.n, 3)
    logw = Vector{T}(undef, s.n)
    lm = Array{T,3}(undef, s.n, 5, s.max_landmarks)
    check(ccall((:slam_pf_download, libslamhip), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.handle, pose, logw, lm))
    pose, logw, lm
end

end # module

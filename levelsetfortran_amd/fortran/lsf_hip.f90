!*************************************************************************************!
! lsf_hip.f90 -- iso_c_binding shim between the reference's Fortran host (set3d.f90) and
! liblsf_hip.so (include/lsf.h).
!
! It gives the host back the procedures of the hot path UNDER THEIR ORIGINAL NAMES AND
! ARGUMENT LISTS, so that the main program calls them exactly as before:
!
!   reinit(phi,gradPhi,gradPhiMag,nx,ny,nz,iter,dx,h)   replaces subs.f90:717-931
!   narrowBand(nx,ny,nz,dx,phi,phiNB,phiSB)             replaces subs.f90:178-207
!   minmaxFlow(phi,phiNB,phiSB,nx,ny,nz,iter,dx,h1)     replaces the loop set3d.f90:394-462
!   phi0Init(phi,nx,ny,nz,dx,xLo,xMin,xMax,surfX,nSurfNode,surfElem,nSurfElem)
!                                                        replaces the loop set3d.f90:218-268
!   advectNodes(phi,phiSB,nx,ny,nz,dx,xLo,surfXX,nSurfNode,iter)
!                                                        replaces set3d.f90:470-479 and :487-501
!   stlRead(surfX,nSurfNode,surfElem,filename,nSurfElem,surfElemTag,surfOrder,nBndComp,nBndElem,bndNormal)
!                                                        replaces subs.f90:17-121 (same list)
!
! and reproduces what the reference prints around them (subs.f90:916,923,929 and
! set3d.f90:449,456,463) and its STOP on a NaN residual (subs.f90:926, set3d.f90:458).
!
! Build with the reference's own flags (-fdefault-real-8, Makefile:4): REAL below is
! then REAL(8) = C double, INTEGER is C int.  INTEGRATION.md shows the two edits to the
! host that bring this module in; levelsetfortran_amd/fortran/Makefile applies them to
! /root/reference/set3d.f90 at build time without copying it into this repository.
!
! Run parameters.  The reference hard-codes them (set3d.f90:140,148,298,390,576) and its
! README announces a namelist ("Working on adding a namelist for inputs", README.md:11);
! this module reads one:
!
!   ./set3d_hip.exec surface.stl [inputs.nml]      (default: ./lsf.nml if it exists)
!
!   &lsf_inputs
!     dx = 0.0234375            ! grid spacing                         (set3d.f90:140)
!     dd = 10                   ! pad cells on every side              (set3d.f90:148)
!     dd_lo = 10, 234, 234      ! pad cells per axis, low / high side  (cubic grids from
!     dd_hi = 10, 235, 235      !   non-cubic bounding boxes)
!     reinit_iter = 10000       ! cap of reinit #1                     (set3d.f90:298)
!     minmax_iter = 200         ! cap of the min/max flow              (set3d.f90:390)
!     reinit2_iter = 2000       ! cap of reinit #2                     (set3d.f90:576)
!     order = 'gs'              ! 'gs' (the reference's raster order, exact) | 'jacobi'
!     arith = 'strict'          ! 'strict' (default: every operation as subs.f90 writes it, the reference's bits) |
!                               ! 'fast' (same mathematics restructured for the GPU, 2 x faster, ~1e-16 per sweep away;
!                               !   see DESIGN.md section 2 for what that becomes over thousands of sweeps)
!     devices = 0, 1, 2, 3      ! reinit runs on these GPUs (lsf_reinit_multi; a device may be listed more than once):
!                               !   order = 'jacobi': block-decomposed, one block each; order = 'gs': ONE GPU -- the process'
!                               !   current device, where the other seams keep their arrays, not the first listed --
!                               !   unless `slabs = 1`; unset: one GPU
!     slabs = 0                 ! 1: with order = 'gs', the reference's ordering over one z slab per listed device, the
!                               !   field of one GPU bit for bit (lsf_reinit_multi with LSF_ORDER_GS).  Opt-in: the path
!                               !   has not run on two real devices yet; on first use every pair of neighbouring devices
!                               !   runs lsf_peer_selftest, and a violated assumption ends the run with its name
!     transport = 'peer'        ! how the blocks exchange their 3-cell halos: 'peer' (peer copies, default) | 'rccl'
!                               !   (ncclSend / ncclRecv over xGMI; needs a distinct device per block)
!     check_every = 8           ! sweeps between two looks of the host at the RMS of a block-decomposed run (1..64;
!                               !   the stop sweep, the field and the printed residuals do not depend on it)
!     resident = 2              ! 0: every seam copies its arrays in and out (default of the C ABI)
!                               ! 1: skip the host-to-device copy of an array the last seam left on the device
!                               ! 2: (default here) ... and leave results on the device until the host needs them:
!                               !    phi, phiNB, phiSB cross PCIe once (lsf_mirror, include/lsf.h)
!   /
!
! Every entry is optional; environment variables of the same meaning (LSF_DX, LSF_DD,
! LSF_DD_{X,Y,Z}_{LO,HI}, LSF_REINIT_ITER, LSF_MINMAX_ITER, LSF_REINIT2_ITER, LSF_ORDER,
! LSF_ARITH, LSF_RESIDENT, LSF_DEVICES="0,1,2,3", LSF_SLABS, LSF_MULTI_TRANSPORT, LSF_MULTI_CHECK_EVERY) override the namelist.
!*************************************************************************************!
MODULE lsf_hip

USE, INTRINSIC :: iso_c_binding
IMPLICIT NONE
PRIVATE
PUBLIC :: reinit, narrowBand, minmaxFlow, phi0Init, advectNodes, lsf_env_real, lsf_env_int, lsf_pad_cells
PUBLIC :: writeVti, snapshotPhi, sumSqDiff, syncHost, syncHostInt, forgetHost, stlRead

INTEGER(c_int), PARAMETER :: LSF_OK = 0, LSF_ERR_NAN = 1
INTEGER(c_int), PARAMETER :: LSF_ORDER_JACOBI = 1, LSF_ARITH_STRICT = 256

! the namelist (read once; unset entries keep these sentinels)
LOGICAL, SAVE :: nml_loaded = .FALSE.
REAL, SAVE :: nml_dx = -1.
INTEGER, SAVE :: nml_dd = -1, nml_dd_lo(3) = -1, nml_dd_hi(3) = -1
INTEGER, SAVE :: nml_reinit_iter = -1, nml_minmax_iter = -1, nml_reinit2_iter = -1
CHARACTER(LEN=16), SAVE :: nml_order = ' ', nml_arith = ' ', nml_transport = ' '
INTEGER, SAVE :: nml_check_every = 8
INTEGER, SAVE :: nml_resident = -1, nml_slabs = 0
INTEGER(c_int), SAVE :: nml_devices(16) = -1
LOGICAL, SAVE :: mirror_set = .FALSE.

INTERFACE
   ! int lsf_reinit(double*,int,int,int,int,double,double,double,int,int*,double*,int)
   FUNCTION lsf_reinit(phi,nx,ny,nz,iter,dx,h,tol,mode,sweeps_done,rms_trace,trace_cap) &
            BIND(C,NAME='lsf_reinit') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(INOUT) :: phi(*)
      INTEGER(c_int), VALUE :: nx,ny,nz,iter,mode,trace_cap
      REAL(c_double), VALUE :: dx,h,tol
      INTEGER(c_int), INTENT(OUT) :: sweeps_done
      REAL(c_double), INTENT(OUT) :: rms_trace(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_reinit
   ! int lsf_reinit_multi(double*,int,int,int,int,double,double,double,int,const int*,int,const int[3],int*,double*,int)
   FUNCTION lsf_reinit_multi(phi,nx,ny,nz,iter,dx,h,tol,mode,devices,ndev,dims,sweeps_done,rms_trace,trace_cap) &
            BIND(C,NAME='lsf_reinit_multi') RESULT(rc)
      IMPORT :: c_int, c_double, c_ptr
      REAL(c_double), INTENT(INOUT) :: phi(*)
      INTEGER(c_int), VALUE :: nx,ny,nz,iter,mode,ndev,trace_cap
      REAL(c_double), VALUE :: dx,h,tol
      INTEGER(c_int), INTENT(IN) :: devices(*)
      TYPE(c_ptr), VALUE :: dims            ! NULL: the library's default decomposition
      INTEGER(c_int), INTENT(OUT) :: sweeps_done
      REAL(c_double), INTENT(OUT) :: rms_trace(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_reinit_multi
   ! int lsf_multi_defaults(int check_every, int transport)   transport: 0 peer copies, 1 RCCL
   FUNCTION lsf_multi_defaults(check_every,transport) BIND(C,NAME='lsf_multi_defaults') RESULT(rc)
      IMPORT :: c_int
      INTEGER(c_int), VALUE :: check_every,transport
      INTEGER(c_int) :: rc
   END FUNCTION lsf_multi_defaults
   FUNCTION lsf_minmax(phi,phiNB,phiSB,nx,ny,nz,iter,dx,h1,tol,mode,iters_done,rms_trace,trace_cap) &
            BIND(C,NAME='lsf_minmax') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(INOUT) :: phi(*)
      INTEGER(c_int), INTENT(INOUT) :: phiNB(*),phiSB(*)
      INTEGER(c_int), VALUE :: nx,ny,nz,iter,mode,trace_cap
      REAL(c_double), VALUE :: dx,h1,tol
      INTEGER(c_int), INTENT(OUT) :: iters_done
      REAL(c_double), INTENT(OUT) :: rms_trace(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_minmax
   FUNCTION lsf_narrowband(phi,phiNB,phiSB,nx,ny,nz,dx) BIND(C,NAME='lsf_narrowband') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(IN) :: phi(*)
      INTEGER(c_int), INTENT(INOUT) :: phiNB(*),phiSB(*)
      INTEGER(c_int), VALUE :: nx,ny,nz
      REAL(c_double), VALUE :: dx
      INTEGER(c_int) :: rc
   END FUNCTION lsf_narrowband
   FUNCTION lsf_phi0(phi,nx,ny,nz,dx,xLo,xMin,xMax,surfX,nSurfNode,surfElem,nSurfElem) &
            BIND(C,NAME='lsf_phi0') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(INOUT) :: phi(*)
      INTEGER(c_int), VALUE :: nx,ny,nz,nSurfNode,nSurfElem
      REAL(c_double), VALUE :: dx
      REAL(c_double), INTENT(IN) :: xLo(3),xMin(3),xMax(3),surfX(*)
      INTEGER(c_int), INTENT(IN) :: surfElem(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_phi0
   FUNCTION lsf_advect_nodes(phi,phiSB,nx,ny,nz,dx,xLo,surfXX,nSurfNode,iters) &
            BIND(C,NAME='lsf_advect_nodes') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(IN) :: phi(*),xLo(3)
      INTEGER(c_int), INTENT(IN) :: phiSB(*)
      INTEGER(c_int), VALUE :: nx,ny,nz,nSurfNode,iters
      REAL(c_double), VALUE :: dx
      REAL(c_double), INTENT(INOUT) :: surfXX(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_advect_nodes
   FUNCTION lsf_mirror(flags) BIND(C,NAME='lsf_mirror') RESULT(rc)
      IMPORT :: c_int
      INTEGER(c_int), VALUE :: flags
      INTEGER(c_int) :: rc
   END FUNCTION lsf_mirror
   FUNCTION lsf_mirror_sync(host) BIND(C,NAME='lsf_mirror_sync') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: host
      INTEGER(c_int) :: rc
   END FUNCTION lsf_mirror_sync
   FUNCTION lsf_mirror_forget(host) BIND(C,NAME='lsf_mirror_forget') RESULT(rc)
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: host
      INTEGER(c_int) :: rc
   END FUNCTION lsf_mirror_forget
   FUNCTION lsf_snapshot(phi,phiO,nx,ny,nz) BIND(C,NAME='lsf_snapshot') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(IN) :: phi(*)
      REAL(c_double), INTENT(INOUT) :: phiO(*)
      INTEGER(c_int), VALUE :: nx,ny,nz
      INTEGER(c_int) :: rc
   END FUNCTION lsf_snapshot
   FUNCTION lsf_sumsq_diff(phi,phiO,nx,ny,nz,s) BIND(C,NAME='lsf_sumsq_diff') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(IN) :: phi(*),phiO(*)
      INTEGER(c_int), VALUE :: nx,ny,nz
      REAL(c_double), INTENT(OUT) :: s
      INTEGER(c_int) :: rc
   END FUNCTION lsf_sumsq_diff
   FUNCTION lsf_write_vti(path,phi,nx,ny,nz,dx,xLo) BIND(C,NAME='lsf_write_vti') RESULT(rc)
      IMPORT :: c_int, c_double, c_char
      CHARACTER(KIND=c_char), INTENT(IN) :: path(*)
      REAL(c_double), INTENT(IN) :: phi(*),xLo(3)
      INTEGER(c_int), VALUE :: nx,ny,nz
      REAL(c_double), VALUE :: dx
      INTEGER(c_int) :: rc
   END FUNCTION lsf_write_vti
   FUNCTION lsf_stl_read(path,nSurfElem,nSurfNode) BIND(C,NAME='lsf_stl_read') RESULT(rc)
      IMPORT :: c_int, c_char
      CHARACTER(KIND=c_char), INTENT(IN) :: path(*)
      INTEGER(c_int), INTENT(OUT) :: nSurfElem,nSurfNode
      INTEGER(c_int) :: rc
   END FUNCTION lsf_stl_read
   FUNCTION lsf_stl_get(surfX,surfElem) BIND(C,NAME='lsf_stl_get') RESULT(rc)
      IMPORT :: c_int, c_double
      REAL(c_double), INTENT(OUT) :: surfX(*)
      INTEGER(c_int), INTENT(OUT) :: surfElem(*)
      INTEGER(c_int) :: rc
   END FUNCTION lsf_stl_get
   FUNCTION lsf_last_error() BIND(C,NAME='lsf_last_error') RESULT(p)
      IMPORT :: c_ptr
      TYPE(c_ptr) :: p
   END FUNCTION lsf_last_error
   FUNCTION c_strlen(s) BIND(C,NAME='strlen') RESULT(n)
      IMPORT :: c_ptr, c_size_t
      TYPE(c_ptr), VALUE :: s
      INTEGER(c_size_t) :: n
   END FUNCTION c_strlen
END INTERFACE

CONTAINS

!*************************************************************************************!
! mode word of include/lsf.h from the environment
!*************************************************************************************!
FUNCTION lsf_mode() RESULT(mode)
INTEGER(c_int) :: mode
CHARACTER(LEN=32) :: v
INTEGER :: st
CALL lsf_load_inputs()
CALL lsf_set_mirror()
mode = 0
v = nml_order
CALL get_environment_variable('LSF_ORDER',v,STATUS=st)
IF (st /= 0) v = nml_order
IF (LEN_TRIM(v) > 0 .AND. TRIM(v) /= 'gs' .AND. TRIM(v) /= 'jacobi') THEN
   PRINT*, " liblsf_hip: order must be 'gs' or 'jacobi', not ",TRIM(v)
   STOP 1
END IF
IF (TRIM(v) == 'jacobi') mode = mode + LSF_ORDER_JACOBI
CALL get_environment_variable('LSF_ARITH',v,STATUS=st)
IF (st /= 0) v = nml_arith
! the reference's own arithmetic unless the run asks for the fast one: a drop-in returns the reference's numbers
IF (LEN_TRIM(v) > 0 .AND. TRIM(v) /= 'strict' .AND. TRIM(v) /= 'fast') THEN
   PRINT*, " liblsf_hip: arith must be 'strict' or 'fast', not ",TRIM(v)
   STOP 1
END IF
IF (TRIM(v) /= 'fast') mode = mode + LSF_ARITH_STRICT
END FUNCTION lsf_mode

!*************************************************************************************!
! &lsf_inputs: from the file named by the second command-line argument, else ./lsf.nml
!*************************************************************************************!
SUBROUTINE lsf_load_inputs()
REAL :: dx
INTEGER :: dd,dd_lo(3),dd_hi(3),reinit_iter,minmax_iter,reinit2_iter,ios,u,resident,devices(16),check_every,slabs
CHARACTER(LEN=16) :: order,arith,transport
CHARACTER(LEN=1024) :: path
LOGICAL :: there
NAMELIST /lsf_inputs/ dx,dd,dd_lo,dd_hi,reinit_iter,minmax_iter,reinit2_iter,order,arith,resident,devices,transport,check_every,slabs
IF (nml_loaded) RETURN
nml_loaded = .TRUE.
path = ' '
IF (command_argument_count() >= 2) CALL get_command_argument(2,path)
IF (LEN_TRIM(path) == 0) path = 'lsf.nml'
INQUIRE(FILE=TRIM(path),EXIST=there)
IF (.NOT. there) THEN
   IF (command_argument_count() >= 2) THEN
      PRINT*, " liblsf_hip: namelist file not found: ",TRIM(path)
      STOP 1
   END IF
   RETURN
END IF
dx = nml_dx; dd = nml_dd; dd_lo = nml_dd_lo; dd_hi = nml_dd_hi
reinit_iter = nml_reinit_iter; minmax_iter = nml_minmax_iter; reinit2_iter = nml_reinit2_iter
order = nml_order; arith = nml_arith; resident = nml_resident; devices = nml_devices
transport = nml_transport; check_every = nml_check_every; slabs = nml_slabs
u = 47
OPEN(UNIT=u,FILE=TRIM(path),STATUS='old',ACTION='read',IOSTAT=ios)
IF (ios == 0) READ(u,NML=lsf_inputs,IOSTAT=ios)
IF (ios /= 0) THEN
   PRINT*, " liblsf_hip: cannot read &lsf_inputs from ",TRIM(path)
   STOP 1
END IF
CLOSE(u)
nml_dx = dx; nml_dd = dd; nml_dd_lo = dd_lo; nml_dd_hi = dd_hi
nml_reinit_iter = reinit_iter; nml_minmax_iter = minmax_iter; nml_reinit2_iter = reinit2_iter
nml_order = order; nml_arith = arith; nml_resident = resident; nml_devices = devices
nml_transport = transport; nml_check_every = check_every; nml_slabs = slabs
PRINT*, " Run parameters read from ",TRIM(path)
END SUBROUTINE lsf_load_inputs

! GPUs for the block-decomposed reinit: LSF_DEVICES="0,1,..." or the namelist's `devices`; nd = leading entries >= 0
SUBROUTINE lsf_device_list(devs,nd)
INTEGER(c_int), INTENT(OUT) :: devs(16),nd
CHARACTER(LEN=256) :: v
INTEGER :: st,ios
CALL lsf_load_inputs()
devs = nml_devices
CALL get_environment_variable('LSF_DEVICES',v,STATUS=st)
IF (st == 0 .AND. LEN_TRIM(v) > 0) THEN
   devs = -1
   v = v(1:LEN_TRIM(v))//' /'                        ! the slash ends the list: entries not given stay -1
   READ(v,*,IOSTAT=ios) devs
   IF (ios /= 0) THEN
      PRINT*, " liblsf_hip: cannot read LSF_DEVICES=",TRIM(v)
      STOP 1
   END IF
END IF
nd = 0
DO WHILE (nd < 16)
   IF (devs(nd+1) < 0) EXIT
   nd = nd+1
END DO
END SUBROUTINE lsf_device_list

SUBROUTINE lsf_fail(where,rc)
CHARACTER(LEN=*), INTENT(IN) :: where
INTEGER(c_int), INTENT(IN) :: rc
TYPE(c_ptr) :: p
CHARACTER(KIND=c_char), POINTER :: s(:)
INTEGER :: n,i
CHARACTER(LEN=512) :: msg
msg = ''
p = lsf_last_error()
IF (c_associated(p)) THEN
   n = MIN(INT(c_strlen(p)),512)
   CALL c_f_pointer(p,s,(/n/))
   DO i = 1,n
      msg(i:i) = s(i)
   END DO
END IF
PRINT*, " liblsf_hip: ",where," failed with code ",rc,": ",TRIM(msg)
STOP 1
END SUBROUTINE lsf_fail

!*************************************************************************************!
! Reinitialize the signed distance function  (same dummy arguments as subs.f90:717-725)
!*************************************************************************************!
SUBROUTINE reinit(phi,gradPhi,gradPhiMag,nx,ny,nz,iter,dx,h)

INTEGER,INTENT(IN) :: nx,ny,nz,iter
REAL,INTENT(IN) :: dx,h
REAL,DIMENSION(0:nx,0:ny,0:nz),INTENT(INOUT) :: phi,gradPhiMag
REAL,DIMENSION(0:nx,0:ny,0:nz,3),INTENT(INOUT) :: gradPhi
REAL,ALLOCATABLE :: trace(:)
INTEGER(c_int) :: rc,done,mode,devs(16),nd
INTEGER :: n,use_slabs

! gradPhi and gradPhiMag are dead outputs of the reference's reinit: the host zeroes them
! right after the first call (set3d.f90:372-375) and never reads them after the second.
! They are left untouched.

ALLOCATE(trace(iter+1))
mode = lsf_mode()
CALL lsf_device_list(devs,nd)
use_slabs = nml_slabs
CALL lsf_env_int('LSF_SLABS',use_slabs)
IF (nd >= 2 .AND. IAND(mode,LSF_ORDER_JACOBI) /= 0) THEN
   ! one block per listed GPU, halos peer to peer, same result as one GPU (include/lsf.h: lsf_reinit_multi)
   PRINT*, " Reinit block-decomposed over ",nd," devices "
   IF (LEN_TRIM(nml_transport) > 0 .AND. TRIM(nml_transport) /= 'peer' .AND. TRIM(nml_transport) /= 'rccl') THEN
      PRINT*, " liblsf_hip: transport must be 'peer' or 'rccl', not ",TRIM(nml_transport)
      STOP 1
   END IF
   rc = lsf_multi_defaults(nml_check_every,MERGE(1,0,TRIM(nml_transport) == 'rccl'))
   IF (rc /= LSF_OK) CALL lsf_fail('lsf_multi_defaults',rc)
   rc = lsf_reinit_multi(phi,nx,ny,nz,iter,dx,h,1.E-5,mode,devs,nd,C_NULL_PTR,done,trace,iter+1)
   IF (rc /= LSF_OK .AND. rc /= LSF_ERR_NAN) CALL lsf_fail('lsf_reinit_multi',rc)
ELSE IF (nd >= 2 .AND. use_slabs /= 0) THEN
   ! the reference's own ordering, one slab of z per listed GPU: the field of one GPU, bit for bit (include/lsf.h).
   ! Opt-in (slabs = 1 / LSF_SLABS=1): without it a device list with order = 'gs' runs on one GPU, as it always did.
   PRINT*, " Reinit in the reference's ordering over ",nd," z slabs "
   rc = lsf_reinit_multi(phi,nx,ny,nz,iter,dx,h,1.E-5,mode,devs,nd,C_NULL_PTR,done,trace,iter+1)
   IF (rc /= LSF_OK .AND. rc /= LSF_ERR_NAN) CALL lsf_fail('lsf_reinit_multi',rc)
ELSE
   IF (nd >= 2) PRINT*, " Reinit in the reference's ordering on one device (slabs = 1 shards it over the listed devices) "
   rc = lsf_reinit(phi,nx,ny,nz,iter,dx,h,1.E-5,mode,done,trace,iter+1)
   IF (rc /= LSF_OK .AND. rc /= LSF_ERR_NAN) CALL lsf_fail('lsf_reinit',rc)
END IF

! what the reference prints, sweep by sweep (subs.f90:915-926)
DO n = 0,done-1
   IF (trace(n+1) < 1.E-5) THEN
      PRINT*, " Distance function time integration has reached steady state "
      EXIT
   END IF
   PRINT*, " Iteration: ",n," ", " RMS Error: ",trace(n+1)
   IF (isnan(trace(n+1))) STOP
END DO
PRINT*
DEALLOCATE(trace)

END SUBROUTINE reinit

!*************************************************************************************!
! Determine Narrow Band  (same dummy arguments as subs.f90:178-184)
!*************************************************************************************!
SUBROUTINE narrowBand(nx,ny,nz,dx,phi,phiNB,phiSB)

INTEGER,INTENT(IN) :: nx,ny,nz
REAL,INTENT(IN) :: dx
REAL,DIMENSION(0:nx,0:ny,0:nz),INTENT(IN) :: phi
INTEGER,DIMENSION(0:nx,0:ny,0:nz),INTENT(INOUT) :: phiNB,phiSB
INTEGER(c_int) :: rc

CALL lsf_set_mirror()
rc = lsf_narrowband(phi,phiNB,phiSB,nx,ny,nz,dx)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_narrowband',rc)

END SUBROUTINE narrowBand

!*************************************************************************************!
! Min/Max Flow: the loop of set3d.f90:394-462 as one call
!*************************************************************************************!
SUBROUTINE minmaxFlow(phi,phiNB,phiSB,nx,ny,nz,iter,dx,h1)

INTEGER,INTENT(IN) :: nx,ny,nz,iter
REAL,INTENT(IN) :: dx,h1
REAL,DIMENSION(0:nx,0:ny,0:nz),INTENT(INOUT) :: phi
INTEGER,DIMENSION(0:nx,0:ny,0:nz),INTENT(INOUT) :: phiNB,phiSB
REAL,ALLOCATABLE :: trace(:)
INTEGER(c_int) :: rc,done
INTEGER :: n

ALLOCATE(trace(MAX(iter,1)))
rc = lsf_minmax(phi,phiNB,phiSB,nx,ny,nz,iter,dx,h1,1.E-7,lsf_mode(),done,trace,MAX(iter,1))
IF (rc /= LSF_OK .AND. rc /= LSF_ERR_NAN) CALL lsf_fail('lsf_minmax',rc)

! what the reference prints (set3d.f90:448-458)
DO n = 1,done
   IF (trace(n) < 1.E-7) THEN
      PRINT*, " Min/max time integration has reached steady state "
      EXIT
   END IF
   PRINT*, " Iteration: ",n," ", " RMS Error: ",trace(n)
   IF (isnan(trace(n))) STOP
END DO
PRINT*
DEALLOCATE(trace)

END SUBROUTINE minmaxFlow

!*************************************************************************************!
! Inside/outside initialisation: the search loop of set3d.f90:218-268 as one call
! (SURVEY.md section 8f rank 1).  phi must hold 1. on entry like at set3d.f90:161.
!*************************************************************************************!
SUBROUTINE phi0Init(phi,nx,ny,nz,dx,xLo,xMin,xMax,surfX,nSurfNode,surfElem,nSurfElem)

INTEGER,INTENT(IN) :: nx,ny,nz
INTEGER*4,INTENT(IN) :: nSurfNode,nSurfElem
REAL,INTENT(IN) :: dx,xLo(3),xMin(3),xMax(3)
REAL,DIMENSION(0:nx,0:ny,0:nz),INTENT(INOUT) :: phi
REAL,INTENT(IN) :: surfX(nSurfNode,3)
INTEGER*4,INTENT(IN) :: surfElem(nSurfElem,3)
INTEGER(c_int) :: rc

CALL lsf_set_mirror()
rc = lsf_phi0(phi,nx,ny,nz,dx,xLo,xMin,xMax,surfX,nSurfNode,surfElem,nSurfElem)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_phi0',rc)

END SUBROUTINE phi0Init

!*************************************************************************************!
! Order-8 gradients on the stencil band + node advection: set3d.f90:470-501 as one call
! (SURVEY.md section 8f rank 3).  surfXX holds the surface nodes on entry (set3d.f90:485).
!*************************************************************************************!
SUBROUTINE advectNodes(phi,phiSB,nx,ny,nz,dx,xLo,surfXX,nSurfNode,iter)

INTEGER,INTENT(IN) :: nx,ny,nz,iter
INTEGER*4,INTENT(IN) :: nSurfNode
REAL,INTENT(IN) :: dx,xLo(3)
REAL,DIMENSION(0:nx,0:ny,0:nz),INTENT(IN) :: phi
INTEGER,DIMENSION(0:nx,0:ny,0:nz),INTENT(IN) :: phiSB
REAL,INTENT(INOUT) :: surfXX(nSurfNode,3)
INTEGER(c_int) :: rc

CALL lsf_set_mirror()
rc = lsf_advect_nodes(phi,phiSB,nx,ny,nz,dx,xLo,surfXX,nSurfNode,iter)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_advect_nodes',rc)

END SUBROUTINE advectNodes

!*************************************************************************************!
! Run-time overrides of the host's hard-coded parameters (set3d.f90:140,148,298,390,576)
! needed by every BASELINE configuration except the as-shipped one.
!*************************************************************************************!
SUBROUTINE lsf_env_real(name,val)
CHARACTER(LEN=*), INTENT(IN) :: name
REAL, INTENT(INOUT) :: val
CHARACTER(LEN=64) :: v
INTEGER :: st,ios
REAL :: t
CALL lsf_load_inputs()
IF (name == 'LSF_DX' .AND. nml_dx > 0.) THEN
   val = nml_dx
   PRINT*, " dx = ",val," (namelist)"
END IF
CALL get_environment_variable(name,v,STATUS=st)
IF (st /= 0) RETURN
READ(v,*,IOSTAT=ios) t
IF (ios == 0) THEN
   val = t
   PRINT*, " ",name," = ",val," (environment override)"
END IF
END SUBROUTINE lsf_env_real

SUBROUTINE lsf_env_int(name,val)
CHARACTER(LEN=*), INTENT(IN) :: name
INTEGER, INTENT(INOUT) :: val
CHARACTER(LEN=64) :: v
INTEGER :: st,ios,t
CALL lsf_load_inputs()
t = -1
IF (name == 'LSF_DD') t = nml_dd
IF (name == 'LSF_REINIT_ITER') t = nml_reinit_iter
IF (name == 'LSF_MINMAX_ITER') t = nml_minmax_iter
IF (name == 'LSF_REINIT2_ITER') t = nml_reinit2_iter
IF (t >= 0) THEN
   val = t
   PRINT*, " ",name(5:)," = ",val," (namelist)"
END IF
CALL get_environment_variable(name,v,STATUS=st)
IF (st /= 0) RETURN
READ(v,*,IOSTAT=ios) t
IF (ios == 0) THEN
   val = t
   PRINT*, " ",name," = ",val," (environment override)"
END IF
END SUBROUTINE lsf_env_int

!*************************************************************************************!
! Pad cells per axis and side (host edit E4b; set3d.f90:148-157 pads every side by dd).
! BASELINE configuration 3 asks for a cubic 512^3 grid around a 12 x 1 x 1 bounding box.
!*************************************************************************************!
SUBROUTINE lsf_pad_cells(dd,ddLo,ddHi)
INTEGER, INTENT(IN) :: dd
INTEGER, INTENT(OUT) :: ddLo(3),ddHi(3)
CHARACTER(LEN=1), PARAMETER :: ax(3) = (/'X','Y','Z'/)
INTEGER :: a
CALL lsf_load_inputs()
ddLo = dd
ddHi = dd
DO a = 1,3
   IF (nml_dd_lo(a) >= 0) ddLo(a) = nml_dd_lo(a)
   IF (nml_dd_hi(a) >= 0) ddHi(a) = nml_dd_hi(a)
   CALL lsf_env_int('LSF_DD_'//ax(a)//'_LO',ddLo(a))
   CALL lsf_env_int('LSF_DD_'//ax(a)//'_HI',ddHi(a))
END DO
END SUBROUTINE lsf_pad_cells

!*************************************************************************************!
! Device-resident chain (include/lsf.h: lsf_mirror).  The host program hands the same
! arrays from seam to seam and never changes them in between; with resident = 2 the
! library keeps them on the device and the four places where the host itself reads phi
! between the seams go through the library as well (host edits E8, E9):
!   snapshotPhi   phiO = phi                        set3d.f90:311
!   sumSqDiff     the asymptotic-error loop         set3d.f90:510-516
!   writeVti      the two VTK writers               set3d.f90:336-351, :554-569
!   syncHost      brings a host array up to date (after the last seam)
!*************************************************************************************!
SUBROUTINE lsf_set_mirror()
INTEGER :: resident,st,ios,t
INTEGER(c_int) :: rc
CHARACTER(LEN=64) :: v
IF (mirror_set) RETURN
mirror_set = .TRUE.
CALL lsf_load_inputs()
resident = 2
IF (nml_resident >= 0) resident = nml_resident
CALL get_environment_variable('LSF_RESIDENT',v,STATUS=st)
IF (st == 0) THEN
   READ(v,*,IOSTAT=ios) t
   IF (ios == 0) resident = t
END IF
rc = 0
IF (resident == 1) rc = lsf_mirror(1_c_int)
IF (resident >= 2) rc = lsf_mirror(3_c_int)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_mirror',rc)
END SUBROUTINE lsf_set_mirror

SUBROUTINE snapshotPhi(phi,phiO,nx,ny,nz)
INTEGER, INTENT(IN) :: nx,ny,nz
REAL, INTENT(IN) :: phi(0:nx,0:ny,0:nz)
REAL, INTENT(INOUT) :: phiO(0:nx,0:ny,0:nz)
INTEGER(c_int) :: rc
CALL lsf_set_mirror()
rc = lsf_snapshot(phi,phiO,nx,ny,nz)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_snapshot',rc)
END SUBROUTINE snapshotPhi

SUBROUTINE sumSqDiff(phi,phiO,nx,ny,nz,s)
INTEGER, INTENT(IN) :: nx,ny,nz
REAL, INTENT(IN) :: phi(0:nx,0:ny,0:nz),phiO(0:nx,0:ny,0:nz)
REAL, INTENT(INOUT) :: s
REAL(c_double) :: t
INTEGER(c_int) :: rc
CALL lsf_set_mirror()
rc = lsf_sumsq_diff(phi,phiO,nx,ny,nz,t)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_sumsq_diff',rc)
s = s + t
END SUBROUTINE sumSqDiff

SUBROUTINE writeVti(name,phi,nx,ny,nz,dx,xLo)
CHARACTER(LEN=*), INTENT(IN) :: name
INTEGER, INTENT(IN) :: nx,ny,nz
REAL, INTENT(IN) :: phi(0:nx,0:ny,0:nz),dx,xLo(3)
INTEGER(c_int) :: rc
CHARACTER(KIND=c_char) :: cname(LEN_TRIM(name)+1)
INTEGER :: i
CALL lsf_set_mirror()
DO i = 1,LEN_TRIM(name)
   cname(i) = name(i:i)
END DO
cname(LEN_TRIM(name)+1) = c_null_char
rc = lsf_write_vti(cname,phi,nx,ny,nz,dx,xLo)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_write_vti',rc)
END SUBROUTINE writeVti

SUBROUTINE syncHost(a)
REAL, INTENT(INOUT), TARGET :: a(*)
INTEGER(c_int) :: rc
rc = lsf_mirror_sync(c_loc(a))
IF (rc /= LSF_OK) CALL lsf_fail('lsf_mirror_sync',rc)
END SUBROUTINE syncHost

! the array is about to be freed (or its device copy is of no further use): drop its twin without a copy
SUBROUTINE forgetHost(a)
REAL, INTENT(INOUT), TARGET :: a(*)
INTEGER(c_int) :: rc
rc = lsf_mirror_forget(c_loc(a))
IF (rc /= LSF_OK) CALL lsf_fail('lsf_mirror_forget',rc)
END SUBROUTINE forgetHost

SUBROUTINE syncHostInt(a)
INTEGER, INTENT(INOUT), TARGET :: a(*)
INTEGER(c_int) :: rc
rc = lsf_mirror_sync(c_loc(a))
IF (rc /= LSF_OK) CALL lsf_fail('lsf_mirror_sync',rc)
END SUBROUTINE syncHostInt

!*************************************************************************************!
! Read STL and Allocate  (same dummy arguments as subs.f90:17; host edit E11)
! The reference merges repeated vertices with a linear search per vertex (quadratic);
! lsf_stl_read returns the same nodes and connectivity from a hash (include/lsf.h).
!*************************************************************************************!
SUBROUTINE stlRead(surfX,nSurfNode,surfElem,filename,nSurfElem,surfElemTag,surfOrder,nBndComp,nBndElem,bndNormal)
CHARACTER, INTENT(IN) :: filename*80
INTEGER*4 nSurfNode,nSurfElem
INTEGER*4,ALLOCATABLE,DIMENSION(:,:),INTENT(OUT) :: surfElem
REAL,ALLOCATABLE,DIMENSION(:,:),INTENT(OUT) :: surfX
INTEGER,ALLOCATABLE,DIMENSION(:),INTENT(OUT) :: surfElemTag,surfOrder
REAL,ALLOCATABLE,DIMENSION(:,:),INTENT(OUT) :: bndNormal
INTEGER,INTENT(OUT) :: nBndComp,nBndElem
INTEGER(c_int) :: rc
CHARACTER(KIND=c_char) :: cname(LEN_TRIM(filename)+1)
INTEGER :: i

PRINT*
PRINT*, " Reading in .stl Mesh "
PRINT*

DO i = 1,LEN_TRIM(filename)
   cname(i) = filename(i:i)
END DO
cname(LEN_TRIM(filename)+1) = c_null_char
rc = lsf_stl_read(cname,nSurfElem,nSurfNode)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_stl_read',rc)
ALLOCATE(surfElem(nSurfElem,3))
ALLOCATE(surfX(nSurfNode,3))
rc = lsf_stl_get(surfX,surfElem)
IF (rc /= LSF_OK) CALL lsf_fail('lsf_stl_get',rc)

! subs.f90:108-118 (the reference allocates bndNormal before it sets nBndComp; one component is what it means)
nBndComp = 1
nBndElem = 0
ALLOCATE(surfOrder(nSurfElem))
ALLOCATE(surfElemTag(nSurfElem))
ALLOCATE(bndNormal(nBndComp,3))
surfOrder = 1
surfElemTag = 0
bndNormal = 0.

END SUBROUTINE stlRead

END MODULE lsf_hip

"""Hand-written GlobalStateParam.txt files with the spellings and layouts a user may produce: input data of
tests/test_config.py::test_parameter_readers_agree_with_the_reference_reader and of make_ref_params.py."""
FILES = {
"spellings": '''currentWorkingDirectory = /no/quotes/here ;
sensorType =2
klgFileName="a b c.klg" ;   // trailing
AssociationFile = "assoc # x.txt" # c
parameterFileCvFormat\t=\t"cam.yaml"\t;
optimizationUseLocalBA = TRUE;
optimizationUseGlobalBA = 0;
preprocessingUsebilateralFilter = yes;
preprocessingInitRadiusMultiplier = 3.5e0;
preprocessingCurvEstimationWindow = .5;
preprocessingCurvValidThreshold = 1e3;
preprocessingNormalEstimationPCA = 0;
preprocessingUseConfEval = 7.9;
preprocessingConfEvalEpsilon = -12.25;
registrationPreAlignSO3 = false ;
registrationJointICPWeight = 100;
registrationICPUseSparseICP = 1;
registrationICPUseCoorespondenceSearch = true;
registrationICPNeighborSearchRadius = -3;
registrationICPUseWeightedICP = False;
registrationICPCurvWeightImpactControl = 5.;
registrationColorUseRGBGrad = 2;
preictionWindowMultiplier = 2;
preictionMinNeighbors = 0x10;
preictionMaxNeighbors = 12abc;
preictionConfThreshold = 2.5f;
fusionCleanWindowMultiplier = 2.25;
globalConfidenceThreshold = 1,5;
globalDenseEnoughThresh = 0.5 0.9;
globalDepthCutoff = +4.0;
globalInputICLNUIMDataset = FALSE;
globalInputLoadTrajectory = "true";
globalInputTrajectoryFormat = 'TUM';
globalInputTrajectoryFile = "a=b.txt";
globalOutputSavePointCloudConfThreshold = 1e-2;
globalStartFrame = 5;
globalEndFrame = 1e2;
globalFrameToSkip = 2;
registrationICPErrorThreshold = 5E-05;
registrationICPCovarianceThreshold = 0.00001;
registrationColorPhotoThreshold = 115.0;
globalOutputSaveTrjectoryFile = true;
globalOutputSaveTrjectoryFileType = "KITTI";
''',
"layout": '''# a
// b
   ## c
sensorType = 1; sensorType = 2;
  klgFileName   =   "x.klg"   ;   
AssociationFile = "";
parameterFileCvFormat = "unterminated
currentWorkingDirectory = "two" "strings";
optimizationUseLocalBA=true
optimizationUseGlobalBA =
preprocessingUsebilateralFilter = true;;
globalDepthCutoff = 3.5;  globalDepthCutoff = 9;
globalDepthCutoff = 2.75 // later wins
GLOBALCONFIDENCETHRESHOLD = 9;
globalConfidenceThreshold= 7 ;
 globalStartFrame = 3
\tglobalEndFrame\t=\t44\t;
globalFrameToSkip = "2";
globalInputTrajectoryFormat = TUM ; // x
globalInputTrajectoryFile = "dir with spaces/traj.txt";  # y
preictionMinNeighbors = 4 ; # z
x
=
a = b = c
registrationJointICPWeight = 10.0;\r
preictionMaxNeighbors = 9\r
''',
}
